// Network engine: RRDBNet (ESRGAN) and SRResNet (SRGAN) forward as a sequence of
// fused conv launches over fp16 blocked-NHWC channel slabs ([C/32][N*H*W][32]) in a caller-provided workspace.
//
// Replaces the nn.Sequential graph of RRDBNet_arch.py:16-62 / SRResNet_arch.py:15-91:
//   torch.cat            -> channel-group offsets inside a 6-group slab (x|x|x1|x2|x3|x4)
//   LeakyReLU / ReLU     -> conv epilogue
//   x5*0.2 + x, RRDB out*0.2 + x, ShortcutBlock x + sub(x) -> conv epilogue
//   Upsample(nearest 2x) -> folded into the next conv's input addressing
//   PixelShuffle(2)      -> folded into the conv's store
#include "common.h"

#include <cmath>
#include <cstring>
#include <mutex>

namespace innfer {

static thread_local std::string g_err;

int set_error(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

namespace {
struct GtRec { std::string name; hipEvent_t e0, e1; double flops, bytes; };
thread_local std::vector<GtRec>* g_gt = nullptr;
thread_local hipEvent_t g_gt_e0 = nullptr;
}  // namespace
bool gt_on() { return g_gt != nullptr; }
void gt_begin(hipStream_t s) {
    if (!g_gt) return;
    g_gt_e0 = nullptr;
    if (hipEventCreate(&g_gt_e0) != hipSuccess || hipEventRecord(g_gt_e0, s) != hipSuccess) g_gt_e0 = nullptr;
}
void gt_end(hipStream_t s, const char* name, double flops, double bytes) {
    if (!g_gt || !g_gt_e0) return;
    hipEvent_t e1 = nullptr;
    if (hipEventCreate(&e1) != hipSuccess || hipEventRecord(e1, s) != hipSuccess) return;
    g_gt->push_back(GtRec{name, g_gt_e0, e1, flops, bytes});
    g_gt_e0 = nullptr;
}

}  // namespace innfer

using namespace innfer;

extern "C" int innfer_timer_start(void) {
    if (g_gt) return set_error(INNFER_ERR_INVALID, "timer_start: a collection is already open on this thread");
    g_gt = new std::vector<GtRec>();
    return INNFER_OK;
}

extern "C" int innfer_timer_stop(void* stream, int cap, char* names, int name_cap, float* ms, double* flops, double* bytes, int* n) {
    if (!g_gt) return set_error(INNFER_ERR_INVALID, "timer_stop: no collection is open on this thread");
    std::vector<GtRec>* recs = g_gt;
    g_gt = nullptr;
    int rc = INNFER_OK;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = set_error(INNFER_ERR_HIP, "timer_stop: stream synchronize failed");
    if (n) *n = (int)recs->size();
    for (int i = 0; i < (int)recs->size(); ++i) {
        GtRec& r = (*recs)[i];
        if (rc == INNFER_OK && i < cap) {
            float t = 0.f;
            (void)hipEventElapsedTime(&t, r.e0, r.e1);
            if (ms) ms[i] = t;
            if (flops) flops[i] = r.flops;
            if (bytes) bytes[i] = r.bytes;
            if (names && name_cap > 0) { strncpy(names + (size_t)i * name_cap, r.name.c_str(), name_cap - 1); names[(size_t)i * name_cap + name_cap - 1] = 0; }
        }
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    delete recs;
    return rc;
}

#ifndef INNFER_ROWP_DEFAULT
#define INNFER_ROWP_DEFAULT 1      // (A/B builds: 0 = the lane-contiguous row order everywhere)
#endif
#ifndef INNFER_FUSE_LAST
#define INNFER_FUSE_LAST 1
#endif
struct ConvSlot {
    std::string key;
    int K = 0, C = 0;
    bool first = false;          // small-Cin VALU conv (fp32 [C*9][K] weights)
    int ksize = 3;               // 1: a 1x1 conv (one-tap panel, ConvLaunch.conv1x1)
    void* d_w = nullptr;         // packed panels (MFMA) or fp32 k-major (first)
    std::vector<float> h_w;      // MFMA convs: the fp32 weights as they were set, kept for the fp32-accurate mode's panels
    void* d_w32 = nullptr;       //   (hi | lo | hi) panels of conv_pack_split, built by innfer_net_set_precision(1)
    bool rowp = false;           // 3x3 convs of exactly 64 outputs: d_w (and d_up4, d_fuse behind it) in the plane row order (ConvLaunch.rowp) -- a store instruction of the 64-channel
                                 // kernels then touches one slab plane; set when the weights are packed.  The split panels (d_w32) keep the lane-contiguous order.
    bool up2x = false;           // the conv of an upconv_block (nearest 2x in front of it): also packed as the four 2x2-tap phases of the equivalent transposed conv
    void* d_up4 = nullptr;       //   conv_pack_deconv2x panels of the summed weights + d_b4 (the bias once per phase), built with d_w
    void* d_up4p = nullptr;      //   the same in the plane row order (rowp slots: the one-visit form, conv3x3_pc UP4); d_up4 stays lane-contiguous (grids <= 16 wide, A/B)
    float* d_b4 = nullptr;
    bool ps2 = false;            // the conv in front of a PixelShuffle(2) stage (nf -> 4 nf, nf 64): also packed phase-major in the plane row order (conv_pack_shuffle2)
    void* d_ps = nullptr;        //   those panels + d_bps (the bias in the same order): the shuffle is the producer / consumer kernel's store (ConvLaunch.rowp = 2)
    float* d_bps = nullptr;
    void* d_fuse = nullptr;      // a last conv of 64 -> <= 3 channels: its panel for the epilogue of HR_conv0 (conv_pack_fuse_last), built with d_w
    float* d_b = nullptr;        // bias padded to the panel width
    bool loaded = false;
    // mode 'NAC' conv blocks (block.py:246-254: norm -> act -> conv): the conv reads act(alpha[c] * x + shift[c]); an elementwise pass in front of it
    bool map_ok = false;         // slot may carry an input map (first conv of an SRResNet block, LR_conv)
    float* d_map = nullptr;      // alpha[C] then shift[C]
    int map_act = 0;
};

struct innfer_net {
    int kind = 0;                // 0 rrdbnet, 1 srresnet
    int in_nc = 3, out_nc = 3, nf = 64, nb = 23, gc = 32, scale = 4, n_up = 2;
    int final_act = 0;           // `finalact` of the reference constructors: activation after the last conv (ConvLaunch.act codes)
    int band_rows = 0;
    int nr = 3;                  // dense blocks per RRDB (RRDBNet_arch.py:73-88)
    int trunk_act = 1;           // `act_type` of the constructors as a ConvLaunch.act code: 1 LeakyReLU(0.2), 2 ReLU
    bool ps_up = false;          // RRDBNet(upsample_mode='pixelshuffle'): conv nf -> 4 nf, PixelShuffle(2), act instead of Upsample, conv, act
    float res_scale = 1.f;       // SRResNet: x + res * res_scale (SRResNet_arch.py:88-91)
    int outm = 0;                // `outm` of RRDBNet / SRResNet.forward (RRDBNet_arch.py:50-62): 0 none, 1 scaltanh, 2 tanh, 3 sigmoid, 4 clamp
    int u8_normalize = 0, u8_round16 = 1;   // innfer_net_forward with INNFER_U8 images: normalize / denormalize flags of np2tensor / tensor2np, fp16 mode
#ifndef INNFER_UP_PHASES_DEFAULT
#define INNFER_UP_PHASES_DEFAULT 1
#endif
    int up_phases = INNFER_UP_PHASES_DEFAULT;   // upconv_block convs as four 2x2-tap phases on the LR grid (innfer_net_set_upconv_phases; the macro: A/B builds)
    int fused_tail = 1;          // HR_conv0 -> conv_last as one kernel where the shapes allow it (innfer_net_set_fused_tail)
#ifndef INNFER_HR_CHAIN_DEFAULT
#define INNFER_HR_CHAIN_DEFAULT 1
#endif
    int hr_chain = INNFER_HR_CHAIN_DEFAULT;   // the LAST upconv_block -> HR_conv0 -> conv_last as one kernel chained through LDS (hr_chain.hip; innfer_net_set_hr_chain): the 64-channel HR tensor is never written
#ifndef INNFER_PS_PC_DEFAULT
#define INNFER_PS_PC_DEFAULT 1
#endif
    int ps_pc = INNFER_PS_PC_DEFAULT;   // PixelShuffle(2) stages on the producer / consumer kernel (phase-major panels); 0 (A/B builds): the two-workgroup kernel of rounds 1-4
    int res_lds = 1;             // the dense block's x5 * 0.2 + x with x taken from the conv's own staged LDS tiles (innfer_net_set_residual_lds; conv3x3_pc RLDS): 1 = where the RRDB's residual follows, 2 = every block
    bool plus = false;           // ESRGAN+ residual paths (RRDBNet_arch.py:155-160)
    int fp32 = 0;                // innfer_net_set_precision: 1 = fp32-accurate forward on split operands (conv3x3.hip SPLIT), the reference's -no_fp16 mode (run.py:345,421-422)
    std::vector<ConvSlot> convs;
};

static int n_upscale(int scale) {
    if (scale == 3) return 1;             // one Upsample(scale_factor=3) stage (RRDBNet_arch.py:33-36)
    int n = 0;
    while ((1 << n) < scale) ++n;
    return n;
}

static void add_conv(innfer_net* net, const std::string& key, int K, int C, bool first = false, int ksize = 3) {
    ConvSlot s;
    s.key = key; s.K = K; s.C = C; s.first = first; s.ksize = ksize;
    net->convs.push_back(s);
}

extern "C" int innfer_version(void) { return INNFER_ABI_VERSION; }
extern "C" const char* innfer_last_error(void) { return g_err.c_str(); }

extern "C" int innfer_rrdbnet_create(innfer_net_t* out, int in_nc, int out_nc, int nf, int nb,
                                     int gc, int scale, int plus) {
    return innfer_rrdbnet_create_ex(out, in_nc, out_nc, nf, nb, gc, scale, plus, 3, 1, 0);
}

extern "C" int innfer_rrdbnet_create_ex(innfer_net_t* out, int in_nc, int out_nc, int nf, int nb,
                                        int gc, int scale, int plus, int nr, int act, int pixelshuffle_up) {
    if (!out) return set_error(INNFER_ERR_INVALID, "rrdbnet_create: null out");
    if (nr < 1 || nr > 64 || (act != 1 && act != 2))
        return set_error(INNFER_ERR_UNSUPPORTED, "rrdbnet_create: nr=%d act=%d (built: nr >= 1; act 1 LeakyReLU(0.2), 2 ReLU)", nr, act);
    if (pixelshuffle_up && scale == 3 && nf != 64)
        return set_error(INNFER_ERR_UNSUPPORTED, "rrdbnet_create: PixelShuffle(3) is built for nf=64 (nf %d: 9 nf is not a multiple of 64)", nf);
    if (scale != 1 && scale != 2 && scale != 3 && scale != 4 && scale != 8 && scale != 16)
        return set_error(INNFER_ERR_UNSUPPORTED, "rrdbnet_create: scale %d (built: 1, 2, 3, 4, 8, 16)", scale);
    if (nf % 32 || gc % 32 || nf <= 0 || gc <= 0 || nf > 64)
        return set_error(INNFER_ERR_UNSUPPORTED, "rrdbnet_create: nf=%d gc=%d (need nf in {32,64}, gc %% 32 == 0)", nf, gc);
    if (in_nc < 1 || in_nc > 8 || out_nc < 1 || out_nc > 16 || nb < 1)
        return set_error(INNFER_ERR_INVALID, "rrdbnet_create: in_nc=%d out_nc=%d nb=%d", in_nc, out_nc, nb);
    innfer_net* net = new innfer_net();
    net->kind = 0; net->in_nc = in_nc; net->out_nc = out_nc; net->nf = nf; net->nb = nb;
    net->gc = gc; net->scale = scale; net->n_up = n_upscale(scale); net->plus = plus != 0;
    net->nr = nr; net->trunk_act = act; net->ps_up = pixelshuffle_up != 0;
    add_conv(net, "model.0", nf, in_nc, true);
    for (int b = 0; b < nb; ++b)
        for (int r = 1; r <= nr; ++r) {
            char rdb[96];        // nr == 3: attributes RDB1..RDB3, else nn.Sequential `RDBs` (RRDBNet_arch.py:73-88)
            if (nr == 3) snprintf(rdb, sizeof rdb, "model.1.sub.%d.RDB%d", b, r);
            else snprintf(rdb, sizeof rdb, "model.1.sub.%d.RDBs.%d", b, r - 1);
            if (plus)            // conv1x1(nf -> gc, no bias), added to x2 (RRDBNet_arch.py:131,155-156)
                add_conv(net, std::string(rdb) + ".conv1x1", gc, nf, false, 1);
            for (int i = 1; i <= 5; ++i)
                add_conv(net, std::string(rdb) + ".conv" + std::to_string(i) + ".0", i < 5 ? gc : nf, nf + (i - 1) * gc);
        }
    add_conv(net, "model.1.sub." + std::to_string(nb), nf, nf);
    net->convs.back().map_ok = true;             // LR_conv under mode 'NAC' with a norm layer (RRDBNet_arch.py:29; block.py:246-254)
    int idx = 2;
    for (int u = 0; u < net->n_up; ++u) {        // upconv_block: Upsample, conv, act -- pixelshuffle_block: conv, PixelShuffle, act (block.py:333-361)
        if (net->ps_up) { add_conv(net, "model." + std::to_string(idx), nf * (scale == 3 ? 9 : 4), nf); net->convs.back().ps2 = scale != 3 && nf == 64; }
        else { add_conv(net, "model." + std::to_string(idx + 1), nf, nf); net->convs.back().up2x = scale != 3; }
        idx += 3;
    }
    add_conv(net, "model." + std::to_string(idx), nf, nf);
    add_conv(net, "model." + std::to_string(idx + 2), out_nc, nf);
    *out = net;
    return INNFER_OK;
}

extern "C" int innfer_srresnet_create(innfer_net_t* out, int in_nc, int out_nc, int nf, int nb, int scale) {
    return innfer_srresnet_create_ex(out, in_nc, out_nc, nf, nb, scale, 2, 1.0f, 0);
}

extern "C" int innfer_srresnet_create_ex(innfer_net_t* out, int in_nc, int out_nc, int nf, int nb, int scale, int act, float res_scale, int upconv_up) {
    if (!out) return set_error(INNFER_ERR_INVALID, "srresnet_create: null out");
    if (scale != 1 && scale != 2 && scale != 3 && scale != 4 && scale != 8)
        return set_error(INNFER_ERR_UNSUPPORTED, "srresnet_create: scale %d (built: 1, 2, 3, 4, 8)", scale);
    if (nf != 64 && nf != 32) return set_error(INNFER_ERR_UNSUPPORTED, "srresnet_create: nf=%d", nf);
    if (scale == 3 && nf != 64 && !upconv_up) return set_error(INNFER_ERR_UNSUPPORTED, "srresnet_create: PixelShuffle(3) is built for nf=64 (9 nf must be a multiple of 64)");
    if (act != 1 && act != 2) return set_error(INNFER_ERR_UNSUPPORTED, "srresnet_create: act %d (1 LeakyReLU(0.2), 2 ReLU)", act);
    if (in_nc < 1 || in_nc > 8 || out_nc < 1 || out_nc > 16 || nb < 1)
        return set_error(INNFER_ERR_INVALID, "srresnet_create: in_nc=%d out_nc=%d nb=%d", in_nc, out_nc, nb);
    innfer_net* net = new innfer_net();
    net->kind = 1; net->in_nc = in_nc; net->out_nc = out_nc; net->nf = nf; net->nb = nb;
    net->gc = 0; net->scale = scale; net->n_up = n_upscale(scale);
    net->trunk_act = act; net->res_scale = res_scale; net->ps_up = !upconv_up;
    add_conv(net, "model.0", nf, in_nc, true);
    for (int b = 0; b < nb; ++b) {
        add_conv(net, "model.1.sub." + std::to_string(b) + ".res.0", nf, nf);
        net->convs.back().map_ok = true;
        add_conv(net, "model.1.sub." + std::to_string(b) + ".res.2", nf, nf);
    }
    add_conv(net, "model.1.sub." + std::to_string(nb), nf, nf);
    net->convs.back().map_ok = true;
    int idx = 2;
    for (int u = 0; u < net->n_up; ++u) {        // pixelshuffle_block: conv, PixelShuffle, act -- upconv_block: Upsample, conv, act (block.py:333-361)
        if (net->ps_up) { add_conv(net, "model." + std::to_string(idx), nf * (scale == 3 ? 9 : 4), nf); net->convs.back().ps2 = scale != 3 && nf == 64; }
        else { add_conv(net, "model." + std::to_string(idx + 1), nf, nf); net->convs.back().up2x = scale != 3; }
        idx += 3;
    }
    add_conv(net, "model." + std::to_string(idx), nf, nf);
    add_conv(net, "model." + std::to_string(idx + 2), out_nc, nf);
    *out = net;
    return INNFER_OK;
}

extern "C" int innfer_net_set_outm(innfer_net_t net, int outm) {
    if (!net || outm < 0 || outm > 4) return set_error(INNFER_ERR_INVALID, "set_outm: 0 none, 1 scaltanh, 2 tanh, 3 sigmoid, 4 clamp");
    net->outm = outm;
    return INNFER_OK;
}

extern "C" void innfer_net_destroy(innfer_net_t net) {
    if (!net) return;
    for (auto& c : net->convs) {
        if (c.d_w) (void)hipFree(c.d_w);
        if (c.d_b) (void)hipFree(c.d_b);
        if (c.d_w32) (void)hipFree(c.d_w32);
        if (c.d_fuse) (void)hipFree(c.d_fuse);
        if (c.d_up4) (void)hipFree(c.d_up4);
        if (c.d_up4p) (void)hipFree(c.d_up4p);
        if (c.d_ps) (void)hipFree(c.d_ps);
        if (c.d_bps) (void)hipFree(c.d_bps);
        if (c.d_b4) (void)hipFree(c.d_b4);
        if (c.d_map) (void)hipFree(c.d_map);
    }
    delete net;
}

extern "C" int innfer_net_num_convs(innfer_net_t net) { return net ? (int)net->convs.size() : INNFER_ERR_INVALID; }
extern "C" int innfer_net_scale(innfer_net_t net) { return net ? net->scale : INNFER_ERR_INVALID; }

extern "C" int innfer_net_conv_info(innfer_net_t net, int idx, char* key, size_t key_cap, int* K, int* C) {
    if (!net || idx < 0 || idx >= (int)net->convs.size()) return set_error(INNFER_ERR_INVALID, "conv_info: bad index %d", idx);
    const ConvSlot& c = net->convs[idx];
    if (key && key_cap) { strncpy(key, c.key.c_str(), key_cap - 1); key[key_cap - 1] = 0; }
    if (K) *K = c.K;
    if (C) *C = c.C;
    return INNFER_OK;
}

// The (wl | wh | wh) panels of one loaded conv for the fp32-accurate mode (conv_pack_split), from the fp32 weights kept at innfer_net_set_conv.
static int build_split_panels(ConvSlot& c) {
    std::vector<char> host(3 * (c.ksize == 1 ? conv_packed_bytes_taps(c.K, c.C, 0x10) : conv_packed_bytes(c.K, c.C)));
    if (c.ksize == 1) conv_pack_1x1_split(c.h_w.data(), c.K, c.C, host.data());
    else conv_pack_split(c.h_w.data(), c.K, c.C, host.data());
    if (hipMalloc(&c.d_w32, host.size()) != hipSuccess) {
        c.d_w32 = nullptr; (void)hipGetLastError();
        return set_error(INNFER_ERR_NOMEM, "out of device memory for the fp32-accurate panels of '%s' (%zu bytes)", c.key.c_str(), host.size());
    }
    if (hipMemcpy(c.d_w32, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(c.d_w32); c.d_w32 = nullptr; (void)hipGetLastError();
        return set_error(INNFER_ERR_HIP, "copying the fp32-accurate panels of '%s' failed", c.key.c_str());
    }
    return INNFER_OK;
}

extern "C" int innfer_net_set_conv(innfer_net_t net, int idx, const float* w, const float* b) {
    if (!net || idx < 0 || idx >= (int)net->convs.size() || !w) return set_error(INNFER_ERR_INVALID, "set_conv: bad arguments");
    ConvSlot& c = net->convs[idx];
    std::vector<char> host;
    size_t bias_n;
    if (c.first) {
        host.resize((size_t)c.C * 9 * c.K * sizeof(float));
        float* d = (float*)host.data();
        for (int ci = 0; ci < c.C; ++ci)
            for (int t = 0; t < 9; ++t)
                for (int k = 0; k < c.K; ++k) d[((size_t)ci * 9 + t) * c.K + k] = w[((size_t)k * c.C + ci) * 9 + t];
        bias_n = c.K;
    } else {
        if (c.ksize == 1) {          // [K,C,1,1]: the one-tap panel of the kernel's 1x1 instantiation
            host.resize(conv_packed_bytes_taps(c.K, c.C, 0x10));
            conv_pack_1x1(w, c.K, c.C, host.data());
        } else {
            host.resize(conv_packed_bytes(c.K, c.C));
            c.rowp = INNFER_ROWP_DEFAULT && c.K == 64;
            conv_pack(w, c.K, c.C, host.data(), c.rowp);
        }
        const int per = 16 * conv_nt_for(c.K);
        bias_n = (size_t)((c.K + per - 1) / per) * per;
        c.h_w.assign(w, w + (size_t)c.K * c.C * c.ksize * c.ksize);
        if (c.d_w32) { (void)hipFree(c.d_w32); c.d_w32 = nullptr; }
        if (c.up2x && c.ksize == 3 && c.K % 64 == 0 && c.C % 32 == 0) {
            // nearest-2x + conv3x3 (block.py:358) == ConvTranspose2d(4, 2, 1) with the taps that meet the same LR pixel summed (conv_pack_up2x_phases)
            std::vector<char> pk(conv_packed_bytes_deconv2x(c.K, c.C));
            conv_pack_up2x_phases(w, c.K, c.C, pk.data());
            std::vector<float> b4((size_t)4 * c.K, 0.f);
            if (b) for (int ph = 0; ph < 4; ++ph) for (int k = 0; k < c.K; ++k) b4[(size_t)ph * c.K + k] = b[k];
            if (!c.d_up4) INNFER_HIP(hipMalloc(&c.d_up4, pk.size()));
            if (!c.d_b4) INNFER_HIP(hipMalloc((void**)&c.d_b4, b4.size() * sizeof(float)));
            INNFER_HIP(hipMemcpy(c.d_up4, pk.data(), pk.size(), hipMemcpyHostToDevice));
            if (c.rowp && c.C == 64) {
                conv_pack_up2x_phases(w, c.K, c.C, pk.data(), 1);
                if (!c.d_up4p) INNFER_HIP(hipMalloc(&c.d_up4p, pk.size()));
                INNFER_HIP(hipMemcpy(c.d_up4p, pk.data(), pk.size(), hipMemcpyHostToDevice));
            }
            INNFER_HIP(hipMemcpy(c.d_b4, b4.data(), b4.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if (c.ps2 && c.ksize == 3 && c.K % 256 == 0 && c.C % 32 == 0) {
            std::vector<char> pk(conv_packed_bytes(c.K, c.C));
            std::vector<float> bp(c.K, 0.f);
            conv_pack_shuffle2(w, b, c.K, c.C, pk.data(), bp.data());
            if (!c.d_ps) INNFER_HIP(hipMalloc(&c.d_ps, pk.size()));
            if (!c.d_bps) INNFER_HIP(hipMalloc((void**)&c.d_bps, bp.size() * sizeof(float)));
            INNFER_HIP(hipMemcpy(c.d_ps, pk.data(), pk.size(), hipMemcpyHostToDevice));
            INNFER_HIP(hipMemcpy(c.d_bps, bp.data(), bp.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if (c.ksize == 3 && c.C == 64 && c.K <= 3) {          // (only the network's last conv has this shape)
            std::vector<char> fp(4096);
            conv_pack_fuse_last(w, c.K, fp.data(), INNFER_ROWP_DEFAULT);          // (fused behind HR_conv0, a 64-output conv: its row order)
            if (!c.d_fuse) INNFER_HIP(hipMalloc(&c.d_fuse, fp.size()));
            INNFER_HIP(hipMemcpy(c.d_fuse, fp.data(), fp.size(), hipMemcpyHostToDevice));
        }
    }
    std::vector<float> bias(bias_n, 0.f);
    if (b) for (int k = 0; k < c.K; ++k) bias[k] = b[k];
    if (!c.d_w) INNFER_HIP(hipMalloc(&c.d_w, host.size()));
    if (!c.d_b) INNFER_HIP(hipMalloc((void**)&c.d_b, bias_n * sizeof(float)));
    INNFER_HIP(hipMemcpy(c.d_w, host.data(), host.size(), hipMemcpyHostToDevice));
    INNFER_HIP(hipMemcpy(c.d_b, bias.data(), bias_n * sizeof(float), hipMemcpyHostToDevice));
    c.loaded = true;
    // a network already in the fp32-accurate mode (set_precision(1) before this set_conv, or one conv re-set afterwards: the ABI-106 call order) gets this
    // conv's split panels here -- set_conv is a load-time call like set_precision; innfer_net_forward never allocates
    if (net->fp32 && !c.first) return build_split_panels(c);
    return INNFER_OK;
}

extern "C" int innfer_net_set_conv_input_map(innfer_net_t net, int idx, const float* alpha, const float* shift, int act) {
    if (!net || idx < 0 || idx >= (int)net->convs.size() || act < 0 || act > 2) return set_error(INNFER_ERR_INVALID, "set_conv_input_map: bad arguments");
    ConvSlot& c = net->convs[idx];
    if (!alpha && !shift && act == 0) {          // no map
        if (c.d_map) { (void)hipFree(c.d_map); c.d_map = nullptr; }
        c.map_act = 0;
        return INNFER_OK;
    }
    if (!c.map_ok)
        return set_error(INNFER_ERR_UNSUPPORTED, "set_conv_input_map: '%s' (built in front of the first conv of an SRResNet block and of LR_conv of either network)", c.key.c_str());
    std::vector<float> h(2 * (size_t)c.C);
    for (int i = 0; i < c.C; ++i) { h[i] = alpha ? alpha[i] : 1.f; h[c.C + i] = shift ? shift[i] : 0.f; }
    if (!c.d_map) INNFER_HIP(hipMalloc((void**)&c.d_map, h.size() * sizeof(float)));
    INNFER_HIP(hipMemcpy(c.d_map, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    c.map_act = act;
    return INNFER_OK;
}

// fp32 = 1: the forward keeps fp32 accuracy -- every activation is a pair of fp16 slabs (hi, lo = (x - hi) * 2^11: 22 significant bits), every weight a
// pair of panels, products are xh wh + 2^-11 (xh wl + xl wh) on the fp16 MFMA path with fp32 accumulation (conv3x3.hip, SPLIT): three times the MFMA
// work and twice the bytes of the fp16 forward.  The reference's fp32 mode on the GPU (`-no_fp16`: run.py:345,421-422).
extern "C" int innfer_net_set_precision(innfer_net_t net, int fp32) {
    if (!net || (fp32 != 0 && fp32 != 1)) return set_error(INNFER_ERR_INVALID, "set_precision: 0 (fp16 arithmetic) or 1 (fp32-accurate)");
    if (fp32) {
        // The (wl | wh | wh) panels of every loaded conv are built HERE -- a load-time call, with its hipMalloc and synchronous copies -- never inside
        // innfer_net_forward (stream capture, concurrent callers).  Idempotent: a conv that already has its panels is skipped; innfer_net_set_conv
        // drops a conv's panels, so the call after a weight change rebuilds exactly those.
        std::vector<ConvSlot*> built;              // panels allocated by THIS call: released again when one of them does not fit
        for (auto& c : net->convs) {
            if (c.first || c.d_w32 || !c.loaded) continue;
            const int rc = build_split_panels(c);
            if (rc != INNFER_OK) {
                for (ConvSlot* q : built) { (void)hipFree(q->d_w32); q->d_w32 = nullptr; }
                return rc;
            }
            built.push_back(&c);
        }
    }
    net->fp32 = fp32;
    return INNFER_OK;
}

extern "C" int innfer_net_set_band_rows(innfer_net_t net, int rows) {
    if (!net || rows < 0) return set_error(INNFER_ERR_INVALID, "set_band_rows: bad arguments");
    net->band_rows = rows;
    return INNFER_OK;
}

extern "C" int innfer_net_set_upconv_phases(innfer_net_t net, int on) {
    if (!net) return set_error(INNFER_ERR_INVALID, "set_upconv_phases: null network");
    if (on < 0 || on > 2) return set_error(INNFER_ERR_INVALID, "set_upconv_phases: 0 (nine taps on the HR grid), 1 (four phases, one visit of a tile: default) or 2 (one phase per visit)");
    net->up_phases = on;
    return INNFER_OK;
}

extern "C" int innfer_net_set_residual_lds(innfer_net_t net, int on) {
    if (!net) return set_error(INNFER_ERR_INVALID, "set_residual_lds: null network");
    if (on < 0 || on > 2) return set_error(INNFER_ERR_INVALID, "set_residual_lds: 0 (epilogue loads), 1 (the RRDB-end layers: default) or 2 (every dense block)");
    net->res_lds = on;
    return INNFER_OK;
}

extern "C" int innfer_net_set_fused_tail(innfer_net_t net, int on) {
    if (!net) return set_error(INNFER_ERR_INVALID, "set_fused_tail: null network");
    net->fused_tail = on ? 1 : 0;
    return INNFER_OK;
}

extern "C" int innfer_net_set_hr_chain(innfer_net_t net, int on) {
    if (!net) return set_error(INNFER_ERR_INVALID, "set_hr_chain: null network");
    net->hr_chain = on ? 1 : 0;
    return INNFER_OK;
}

extern "C" int innfer_net_set_u8_io(innfer_net_t net, int normalize, int fp16_mode) {
    if (!net) return set_error(INNFER_ERR_INVALID, "set_u8_io: null net");
    net->u8_normalize = normalize != 0;
    net->u8_round16 = fp16_mode != 0;
    return INNFER_OK;
}

extern "C" int innfer_net_set_final_act(innfer_net_t net, int act) {
    if (!net || (act != 0 && act != 1 && act != 2 && act != 3 && act != 6))
        return set_error(INNFER_ERR_INVALID, "set_final_act: act %d (0 none, 1 LeakyReLU(0.2), 2 ReLU, 3 tanh, 6 sigmoid)", act);
    net->final_act = act;
    return INNFER_OK;
}

extern "C" double innfer_net_flops(innfer_net_t net, int N, int H, int W) {
    if (!net) return 0.0;
    double px = (double)N * H * W, f = 0.0;
    const int nconv = (int)net->convs.size();
    // resolution multiplier per conv: trunk at 1x, up-conv u at 4^(u+1) (RRDB: conv runs AFTER the
    // upsample; SRGAN: conv runs BEFORE the shuffle, i.e. at 4^u), tail convs at scale^2.
    for (int i = 0; i < nconv; ++i) {
        const ConvSlot& c = net->convs[i];
        double mult = 1.0;
        const int tail0 = nconv - 2 - net->n_up;          // first up conv
        if (i >= tail0) {
            const int u = i - tail0;
            const double f2 = net->scale == 3 ? 9.0 : 4.0;     // pixels per input pixel after one upsample stage
            if (u < net->n_up) mult = std::pow(f2, !net->ps_up ? u + 1 : u);
            else mult = std::pow(f2, net->n_up);
        }
        f += 2.0 * c.ksize * c.ksize * c.K * c.C * px * mult;
    }
    return f;
}

// Workspace carve (all fp16 NHWC):
//   fea        [N,H,W,nf]               conv_first output, kept for the trunk shortcut
//   slab[3]    [N,H,W,nf+4gc]           rotating RDB slabs (RRDB) / ping-pong features (SRGAN)
//   trunk      [N,H,W,nf]
//   up[u]      [N,2^(u+1)H,2^(u+1)W,nf] after each upsample stage
//   hr         [N,sH,sW,nf]             HR_conv0 output
struct Carve {
    size_t fea, slab[3], trunk, up[5], hr, tmp, total;
    size_t lo;                   // fp32-accurate mode: every buffer's lo twin lies `lo` bytes behind it (the second half of the workspace)
    int slab_w;
};

static Carve carve(const innfer_net* net, int N, int H, int W) {
    Carve c{};
    const size_t px = (size_t)N * H * W;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    c.slab_w = net->kind == 0 ? net->nf + 4 * net->gc : net->nf;
    size_t off = 0;
    c.fea = off; off += al(px * net->nf * 2);
    for (int i = 0; i < 3; ++i) { c.slab[i] = off; off += al(px * c.slab_w * 2); }
    c.trunk = off; off += al(px * net->nf * 2);
    c.tmp = off; if (net->plus) off += al(px * net->gc * 2);          // conv1x1(x) of the current RDB
    size_t m = 1;
    for (int u = 0; u < net->n_up; ++u) { m *= net->scale == 3 ? 9 : 4; c.up[u] = off; off += al(px * m * net->nf * 2); }
    c.hr = off; off += al(px * m * net->nf * 2);
    c.lo = net->fp32 ? off : 0;
    c.total = net->fp32 ? 2 * off : off;
    return c;
}

extern "C" size_t innfer_net_workspace_bytes(innfer_net_t net, int N, int H, int W) {
    if (!net || N <= 0 || H <= 0 || W <= 0) return 0;
    return carve(net, N, H, W).total;
}

namespace {

// Optional per-launch timing (innfer_net_forward_timed): HIP events on the launch stream
// bracket every kernel of one forward.
struct LaunchTimer {
    std::vector<hipEvent_t> ev;      // 2 per launch
    std::vector<double> flops;
    std::vector<double> bytes;       // algorithmic HBM bytes of the launch: every operand read once, every result written once
    std::vector<int> kind;           // 0 first conv; 16*NT + out_mode (+ 1000 / 2000 / 3000: see do_conv) for the MFMA convs
};
thread_local LaunchTimer* g_timer = nullptr;

int timed_begin(hipStream_t s) {
    if (!g_timer) return INNFER_OK;
    hipEvent_t e;
    INNFER_HIP(hipEventCreate(&e));
    INNFER_HIP(hipEventRecord(e, s));
    g_timer->ev.push_back(e);
    return INNFER_OK;
}

int timed_end(hipStream_t s, double flops, double bytes, int kind) {
    if (!g_timer) return INNFER_OK;
    hipEvent_t e;
    INNFER_HIP(hipEventCreate(&e));
    INNFER_HIP(hipEventRecord(e, s));
    g_timer->ev.push_back(e);
    g_timer->flops.push_back(flops);
    g_timer->bytes.push_back(bytes);
    g_timer->kind.push_back(kind);
    return INNFER_OK;
}

// INNFER_DEBUG=1: print every launch and synchronise after it (fault localisation only).
bool debug_sync() {
    static const bool on = getenv("INNFER_DEBUG") != nullptr;
    return on;
}

int debug_after(const char* what, hipStream_t s) {
    if (!debug_sync()) return INNFER_OK;
    fprintf(stderr, "[innfer] %s ... ", what); fflush(stderr);
    hipError_t e = hipStreamSynchronize(s);
    fprintf(stderr, "%s\n", hipGetErrorString(e)); fflush(stderr);
    return e == hipSuccess ? INNFER_OK : set_error(INNFER_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
}

int do_conv(const ConvLaunch& L, hipStream_t s) {
    int rc = timed_begin(s);
    if (rc) return rc;
    if (debug_sync())
        fprintf(stderr, "[innfer] conv C=%d K=%d N=%d H=%d W=%d up=%d act=%d mode=%d rows=[%d,%d) in_g=%ld out_g=%ld\n",
                L.C, L.K, L.N, L.H, L.W, L.up, L.act, L.out_mode, L.y0, L.y1, L.in_gstride, L.out_gstride);
    {
        // whole-frame launches alternate their traversal direction (Infinity Cache reuse between layers)
        const bool alt = INNFER_KNOB("INNFER_TILE_REV", 1) != 0;
        thread_local unsigned parity = 0;
        ConvLaunch R = L;
        const int y1 = L.y1 > 0 ? L.y1 : L.H;
        R.rev = (alt && L.y0 == 0 && y1 == L.H) ? (int)(parity++ & 1) : 0;
        rc = conv_launch(R, s);
    }
    if (rc) return rc;
    rc = debug_after("conv3x3", s);
    if (rc) return rc;
    const int y1 = L.y1 > 0 ? L.y1 : L.H;
    const double px = (double)L.N * (y1 - L.y0) * L.W;
    const int taps = L.conv1x1 ? 1 : 9;
    // algorithmic bytes per output pixel: C input channels (a quarter of them per pixel behind the folded nearest-2x), K outputs, K per residual
    const double obytes = L.out_mode == OUT_NCHW ? (L.out_u8 ? 1.0 : L.out_f32 ? 4.0 : 2.0) : 2.0;
    const double bytes = px * (L.C * 2.0 / (L.up ? 4.0 : 1.0) + L.K * obytes + (L.res1 ? L.K * 2.0 : 0.0) + (L.res2 ? L.K * 2.0 : 0.0))
                         + (double)taps * L.K * L.C * 2.0;
    if (L.fuse_w) {      // HR_conv0 + conv_last as one launch (+ the rim pass): both convs' FLOPs; bytes: C channels in, the planar result out, both weight sets
        const double ob = L.fuse_out_mode == 2 ? 1.0 : L.fuse_out_mode == 1 ? 4.0 : 2.0;
        return timed_end(s, 2.0 * 9.0 * (L.K * (double)L.C + 64.0 * L.fuse_oc) * px, px * (L.C * 2.0 + L.fuse_oc * ob) + 9.0 * (L.K * L.C + 64.0 * L.fuse_oc) * 2.0, 2000 + 16 * conv_nt_for(L.K) + L.out_mode);
    }
    // launch kinds: 16 * NT + out_mode, + 1000 fp32-accurate (split) form, + 2000 fused HR_conv0 + conv_last (TMF 0x201FF), + 3000 the four 2x2-tap phases
    // of an upconv_block (TM 0x1B, K = 4 * phase_c on the LR grid), + 4000 the PixelShuffle(2) store on the producer / consumer kernel (TMF 0xC001FF): their own
    // instantiations, their own rows in the bench's per-kernel table.
    // The phase form's FLOPs are the ALGORITHMIC ones of the layer it replaces (nine taps on the 2H x 2W grid = 2 * 9 * 4K * C per LR pixel); it executes 4/9 of them.
    // (fp32-accurate mode: the same algorithmic FLOPs -- executed: 3x --, two slabs per tensor, three panels per weight)
    return timed_end(s, 2.0 * taps * L.K * L.C * px, L.split ? 2.0 * bytes + (double)taps * L.K * L.C * 2.0 : bytes,
                     16 * conv_nt_for(L.K) + L.out_mode + (L.split ? 1000 : 0) + (L.deconv_phases ? 3000 : 0) + ((L.out_mode == OUT_SHUFFLE2 && L.rowp == 2) ? 4000 : 0));
}

// nearest-neighbour upsampling of a slab by an integer factor (src = dst / f, block.py:321-322).  The 2x case is folded into the conv's input
// addressing; the 3x models (one stage, RRDBNet_arch.py:33-36) materialise the tensor: a tile's source columns are not a fixed function of
// the column inside the tile when the tile width (32) is not a multiple of the factor.  One thread per (output pixel, 16-byte piece).
__global__ void slab_upsample_nearest(const f16* src, long src_g, f16* dst, long dst_g, int groups, int N, int H, int W, int f) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int HO = H * f, WO = W * f;
    const long M = (long)N * HO * WO;
    if (i >= M * 4 * groups) return;
    const int q = (int)(i & 3);
    const long r = i >> 2;
    const int g = (int)(r / M);
    const long m = r - g * M;
    const int X = (int)(m % WO), Y = (int)((m / WO) % HO);
    const long n = m / ((long)WO * HO);
    const long sp = (n * H + Y / f) * W + X / f;
    *(f16x8*)(dst + g * dst_g + m * 32 + q * 8) = *(const f16x8*)(src + g * src_g + sp * 32 + q * 8);
}

// Input map of a 'NAC' conv block: dst = act(alpha[c] * src + shift[c]) on a slab (8 channels per thread)
__global__ void slab_input_map(const f16* src, f16* dst, long g_elems, int groups, const float* map, int C, int act) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = g_elems / 8;
    if (i >= per * groups) return;
    const int g = (int)(i / per);
    const long r = i % per;
    const int c = g * 32 + (int)(r & 3) * 8;
    const f16x8 x = *(const f16x8*)(src + g * g_elems + r * 8);
    f16x8 y;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = c + e < C ? (float)x[e] * map[c + e] + map[C + c + e] : 0.f;
        if (act == 1) v = fmaxf(v, 0.2f * v);
        else if (act == 2) v = fmaxf(v, 0.f);
        y[e] = (f16)v;
    }
    *(f16x8*)(dst + g * g_elems + r * 8) = y;
}

// The same on the (hi, lo) slab pairs of the fp32-accurate mode: x = hi + lo * 2^-11, mapped in fp32, split again
__global__ void slab_input_map_split(const f16* src, f16* dst, long lo, long g_elems, int groups, const float* map, int C, int act) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = g_elems / 8;
    if (i >= per * groups) return;
    const int g = (int)(i / per);
    const long r = i % per;
    const int c = g * 32 + (int)(r & 3) * 8;
    const f16x8 xh = *(const f16x8*)(src + g * g_elems + r * 8), xl = *(const f16x8*)(src + lo + g * g_elems + r * 8);
    f16x8 yh, yl;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = __builtin_fmaf((float)xl[e], 1.0f / 2048.0f, (float)xh[e]);
        float v = c + e < C ? x * map[c + e] + map[C + c + e] : 0.f;
        if (act == 1) v = fmaxf(v, 0.2f * v);
        else if (act == 2) v = fmaxf(v, 0.f);
        yh[e] = (f16)v;
        yl[e] = (f16)((v - (float)yh[e]) * 2048.0f);
    }
    *(f16x8*)(dst + g * g_elems + r * 8) = yh;
    *(f16x8*)(dst + lo + g * g_elems + r * 8) = yl;
}

// PixelShuffle(r) between slabs: out[n, y r + i, x r + j, c] = in[n, y, x, c r^2 + i r + j] (torch.nn.PixelShuffle); 8 output channels per thread
__global__ void slab_pixel_shuffle(const f16* src, long src_g, f16* dst, long dst_g, int nf, int N, int H, int W, int r) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = nf / 8, WO = W * r, HO = H * r;
    if (i >= (long)N * HO * WO * c8) return;
    const int c = (int)(i % c8) * 8;
    const long m = i / c8;
    const int X = (int)(m % WO), Y = (int)((m / WO) % HO);
    const long n = m / ((long)WO * HO);
    const long sp = (n * H + Y / r) * W + X / r;
    const int sub = (Y % r) * r + X % r;
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = (c + e) * r * r + sub;
        v[e] = src[(long)(ch >> 5) * src_g + sp * 32 + (ch & 31)];
    }
    *(f16x8*)(dst + (long)(c >> 5) * dst_g + m * 32 + (c & 31)) = v;
}

int do_input_map(const ConvSlot& cs, const f16* src, f16* dst, long G, hipStream_t s, long lo = 0) {
    const int groups = (cs.C + 31) / 32;
    const long n = G / 8 * groups;
    if (lo) hipLaunchKernelGGL(slab_input_map_split, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, lo, G, groups, (const float*)cs.d_map, cs.C, cs.map_act);
    else hipLaunchKernelGGL(slab_input_map, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, G, groups, (const float*)cs.d_map, cs.C, cs.map_act);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

int do_first(const FirstConvLaunch& F, hipStream_t s) {
    int rc = timed_begin(s);
    if (rc) return rc;
    rc = first_conv_launch(F, s);
    if (rc) return rc;
    rc = debug_after("first_conv", s);
    if (rc) return rc;
    const double px = (double)F.N * F.H * F.W;
    return timed_end(s, 2.0 * 9.0 * F.K * F.Cin * px, px * (F.Cin * (F.in_f32 ? 4.0 : 2.0) + F.K * 2.0 * (F.out2 ? 2 : 1)), 0);
}

struct Plan {                    // one MFMA conv in the launch list
    ConvLaunch L;
};

// in/out point at channel 0 of the tensors; *_g are their group strides (elements).
// lo > 0: fp32-accurate mode, every slab's lo twin `lo` elements behind it
ConvLaunch mk_launch(const ConvSlot& cs, const f16* in, long in_g, void* out, long out_g,
                     int N, int H, int W, int act, long lo) {
    ConvLaunch L{};
    L.in = in; L.in_gstride = in_g; L.C = cs.C;
    L.wpk = (const f16*)(lo ? cs.d_w32 : cs.d_w); L.bias = cs.d_b;
    if (lo) { L.split = 1; L.in_lo = L.out_lo = L.res1_lo = L.res2_lo = lo; }
    L.out = out; L.out_gstride = out_g; L.out_coff = 0; L.K = cs.K;
    L.N = N; L.H = H; L.W = W; L.act = act;
    L.s1 = 1.f; L.s2 = 1.f;
    L.y0 = 0; L.y1 = H;
    L.out_mode = OUT_SLAB;
    L.conv1x1 = cs.ksize == 1;
    L.rowp = (!lo && cs.rowp) ? 1 : 0;          // (the split panels keep the lane-contiguous order)
    return L;
}

// Launch a dependent chain of same-resolution convs.  band_rows == 0: one launch per
// layer over the whole frame.  band_rows > 0: skewed row bands -- layer l of band b
// covers rows [b*R - l, (b+1)*R - l), so every row a layer reads from its predecessor
// (its own rows +-1) has already been produced, nothing is recomputed, and the ~R-row
// working set of consecutive layers stays resident in the 256 MiB Infinity Cache
// instead of streaming 1664 B/pixel/RDB from HBM.
int run_chain(std::vector<ConvLaunch>& chain, int band_rows, hipStream_t s) {
    if (chain.empty()) return INNFER_OK;
    const int H = chain[0].H, L = (int)chain.size();
    if (band_rows <= 0 || band_rows >= H) {
        for (auto& c : chain) { int rc = do_conv(c, s); if (rc) return rc; }
        return INNFER_OK;
    }
    const int nbands = (H + (L - 1) + band_rows - 1) / band_rows;
    for (int b = 0; b < nbands; ++b)
        for (int l = 0; l < L; ++l) {
            int y0 = b * band_rows - l, y1 = y0 + band_rows;
            if (y0 < 0) y0 = 0;
            if (y1 > H) y1 = H;
            if (b == nbands - 1) y1 = H;
            if (y0 >= y1) continue;
            ConvLaunch c = chain[l];
            c.y0 = y0; c.y1 = y1;
            int rc = do_conv(c, s);
            if (rc) return rc;
        }
    return INNFER_OK;
}

}  // namespace

extern "C" int innfer_net_forward(innfer_net_t net, const void* d_in, int in_dtype, void* d_out, int out_dtype,
                                  int N, int H, int W, void* d_ws, size_t ws_bytes, void* stream) {
    if (!net || !d_in || !d_out || !d_ws) return set_error(INNFER_ERR_INVALID, "forward: null argument");
    if (N <= 0 || H <= 0 || W <= 0) return set_error(INNFER_ERR_INVALID, "forward: bad shape %dx%dx%d", N, H, W);
    auto dtype_ok = [](int d) { return d == INNFER_F16 || d == INNFER_F32 || d == INNFER_U8; };
    if (!dtype_ok(in_dtype) || !dtype_ok(out_dtype)) return set_error(INNFER_ERR_INVALID, "forward: bad dtype");
    if (out_dtype == INNFER_U8 && net->out_nc > 4) return set_error(INNFER_ERR_UNSUPPORTED, "forward: a uint8 image has at most 4 channels (out_nc %d)", net->out_nc);
    for (auto& c : net->convs)
        if (!c.loaded) return set_error(INNFER_ERR_INVALID, "forward: weights of '%s' were never set", c.key.c_str());
    const Carve cv = carve(net, N, H, W);
    if (ws_bytes < cv.total) return set_error(INNFER_ERR_WORKSPACE, "forward: workspace %zu < %zu bytes", ws_bytes, cv.total);
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)d_ws;
    const int nf = net->nf, gc = net->gc;
    const long G = (long)N * H * W * 32;          // group stride of every LR-resolution slab
    const long LO = (long)(cv.lo / 2);            // fp32-accurate mode: elements from a slab to its lo twin (0: fp16 mode)
    if (net->fp32) {
        if (in_dtype == INNFER_F16) return set_error(INNFER_ERR_INVALID, "forward: the fp32-accurate mode takes fp32 or uint8 input (an fp16 tensor is the fp16 mode's)");
        for (auto& c : net->convs)                // built by innfer_net_set_precision(1) (load time); a conv re-set since then rebuilds there too
            if (!c.first && !c.d_w32) return set_error(INNFER_ERR_INVALID, "forward: the fp32-accurate panels of '%s' are missing: call innfer_net_set_precision(net, 1) after the last innfer_net_set_conv", c.key.c_str());
    }
    auto mk = [&](const ConvSlot& cs, const f16* in, long in_g, void* out, long out_g, int N_, int H_, int W_, int act) {
        return mk_launch(cs, in, in_g, out, out_g, N_, H_, W_, act, LO);
    };
    f16* fea = (f16*)(ws + cv.fea);
    f16* slab[3] = {(f16*)(ws + cv.slab[0]), (f16*)(ws + cv.slab[1]), (f16*)(ws + cv.slab[2])};
    f16* trunk = (f16*)(ws + cv.trunk);
    int ci = 0;

    {   // conv_first: NCHW input -> fea (+ slab[0][0:nf))
        const ConvSlot& c0 = net->convs[ci++];
        FirstConvLaunch F{};
        F.in = d_in; F.in_f32 = in_dtype == INNFER_F32; F.Cin = c0.C; F.w = (const float*)c0.d_w; F.bias = c0.d_b;
        F.in_u8 = in_dtype == INNFER_U8; F.in_norm = net->u8_normalize; F.in_round16 = net->u8_round16;     // np2tensor as the conv's prologue
        F.out = fea; F.out_gstride = G; F.out2 = slab[0]; F.out2_gstride = G;
        // (SRResNet: the first block reads fea itself -- a second copy in slab[0] only serves the dense blocks' concatenation; one 128 B / pixel store less in a
        //  launch that is bound by its stores)
        if (net->kind == 1 && net->nb > 0) F.out2 = nullptr;
        F.K = nf; F.N = N; F.H = H; F.W = W; F.act = 0;
        F.out_lo = LO; F.out2_lo = LO;
        int rc = do_first(F, s);
        if (rc) return rc;
    }

    std::vector<ConvLaunch> chain;
    int cur = 0;
    if (net->kind == 0) {
        for (int b = 0; b < net->nb; ++b) {
            // three rotating slabs: the RRDB's input stays untouched (its x*0.2 + x residual) while the blocks ping-pong between the other two
            const int in_slab = cur;
            const int f1 = (cur + 1) % 3, f2 = (cur + 2) % 3;
            const int nr = net->nr;
            int last = in_slab;
            for (int r = 0; r < nr; ++r) {
                const int work_r = last, dest_r = (r == 0 || last == f2) ? f1 : f2;
                last = dest_r;
                f16* S = slab[work_r];
                f16* t1x1 = (f16*)(ws + cv.tmp);
                if (net->plus) {                 // t = conv1x1(x): no bias, no activation
                    const ConvSlot& cs = net->convs[ci++];
                    chain.push_back(mk(cs, S, G, t1x1, G, N, H, W, 0));
                }
                for (int i = 0; i < 4; ++i) {
                    const ConvSlot& cs = net->convs[ci++];
                    ConvLaunch L = mk(cs, S, G, S + (long)((nf + i * gc) / 32) * G, G, N, H, W, net->trunk_act);
                    if (net->plus && i == 1) { L.res1 = t1x1; L.res1_gstride = G; L.s1 = 1.f; }          // x2 += conv1x1(x)
                    if (net->plus && i == 3) {                                                             // x4 += x2
                        L.res1 = S + (long)((nf + gc) / 32) * G; L.res1_gstride = G; L.s1 = 1.f;
                    }
                    chain.push_back(L);
                }
                const ConvSlot& cs = net->convs[ci++];
                ConvLaunch L = mk(cs, S, G, slab[dest_r], G, N, H, W, 0);
                L.res1 = S; L.res1_gstride = G; L.s1 = 0.2f;                      // x5*0.2 + x
                L.res1_lds = net->res_lds;                                        // ... with x from the conv's own LDS stages where the kernel has that form (fp16 engine, nf 64)
                if (r == nr - 1) { L.res2 = slab[in_slab]; L.res2_gstride = G; L.s2 = 0.2f; }   // RRDB: out*0.2 + x
                chain.push_back(L);
            }
            cur = last;
        }
    } else {
        // SRGAN residual blocks: t = t + conv(relu(conv(t))) on nf-wide slabs
        for (int b = 0; b < net->nb; ++b) {
            const int a = cur, m = (cur + 1) % 3, o = (cur + 2) % 3;
            const ConvSlot& c0 = net->convs[ci++];
            const f16* xa = b == 0 ? fea : slab[a];                  // the block's input (block 0: conv_first's output, never rewritten: the last shortcut reads it too)
            const f16* in0 = xa;
            if (c0.d_map) {          // mode 'NAC': norm -> act in front of the conv, into the (still unused) trunk slab; launches in order, whole frame
                int rc = run_chain(chain, 0, s);
                if (rc) return rc;
                chain.clear();
                rc = do_input_map(c0, xa, trunk, G, s, LO);
                if (rc) return rc;
                in0 = trunk;
            }
            chain.push_back(mk(c0, in0, G, slab[m], G, N, H, W, net->trunk_act));
            const ConvSlot& c1 = net->convs[ci++];
            ConvLaunch L = mk(c1, slab[m], G, slab[o], G, N, H, W, 0);
            L.res1 = xa; L.res1_gstride = G; L.s1 = net->res_scale;               // x + res * res_scale
            chain.push_back(L);
            cur = o;
        }
    }
    {   // trunk conv + ShortcutBlock: fea + conv(t)
        const ConvSlot& cs = net->convs[ci++];
        const f16* in = slab[cur];
        if (cs.d_map) {
            int rc = run_chain(chain, 0, s);
            if (rc) return rc;
            chain.clear();
            rc = do_input_map(cs, slab[cur], slab[(cur + 1) % 3], G, s, LO);
            if (rc) return rc;
            in = slab[(cur + 1) % 3];
        }
        ConvLaunch L = mk(cs, in, G, trunk, G, N, H, W, 0);
        L.res1 = fea; L.res1_gstride = G; L.s1 = 1.f;
        chain.push_back(L);
    }
    bool any_map = false;
    for (auto& c : net->convs) any_map |= c.d_map != nullptr;
    int rc = run_chain(chain, (any_map || net->fp32) ? 0 : net->band_rows, s);
    if (rc) return rc;

    const f16* t = trunk;
    int h = H, w = W;
    for (int u = 0; u < net->n_up; ++u) {
        const ConvSlot& cs = net->convs[ci++];
        f16* dst = (f16*)(ws + cv.up[u]);
        const long gi = (long)N * h * w * 32, go = gi * 4;
        if (net->ps_up && (net->scale == 3 || net->nf != 64 || net->fp32)) {
            // PixelShuffle(3), or factor 2 on 32 features: the conv (nf -> r^2 nf, act in its epilogue -- it commutes with the permutation) writes a
            // slab in the (still unused) HR region, one gather pass rearranges it (block.py:333-346).  The 64-feature factor-2 stages shuffle in the
            // conv's own store (OUT_SHUFFLE2) below.
            const int r = net->scale == 3 ? 3 : 2;
            f16* Y = (f16*)(ws + cv.hr);
            rc = do_conv(mk(cs, t, gi, Y, gi, N, h, w, net->trunk_act), s);
            if (rc) return rc;
            const long gr = gi * r * r, nthr = (long)N * h * w * r * r * (net->nf / 8);
            hipLaunchKernelGGL(slab_pixel_shuffle, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, (const f16*)Y, gi, dst, gr, net->nf, N, h, w, r);
            if (LO) hipLaunchKernelGGL(slab_pixel_shuffle, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, (const f16*)Y + LO, gi, dst + LO, gr, net->nf, N, h, w, r);
            INNFER_HIP(hipGetLastError());
            t = dst; h *= r; w *= r;
            continue;
        }
        if (net->scale == 3) {      // Upsample(nearest 3x), materialised in the (still unused) HR slab -> conv -> LeakyReLU
            f16* U = (f16*)(ws + cv.hr);
            const long g3 = gi * 9, nthr = (long)N * 9 * h * w * 4 * (net->nf / 32);
            hipLaunchKernelGGL(slab_upsample_nearest, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, t, gi, U, g3, net->nf / 32, N, h, w, 3);
            if (LO) hipLaunchKernelGGL(slab_upsample_nearest, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, t + LO, gi, U + LO, g3, net->nf / 32, N, h, w, 3);
            INNFER_HIP(hipGetLastError());
            rc = do_conv(mk(cs, U, g3, dst, g3, N, 3 * h, 3 * w, net->trunk_act), s);
            if (rc) return rc;
            t = dst; h *= 3; w *= 3;
            continue;
        }
        if (u == net->n_up - 1 && !net->ps_up && net->up_phases == 1 && net->hr_chain && !net->fp32 && cs.d_up4p && !cs.d_map && (net->trunk_act == 1 || net->trunk_act == 2) && w > 16 &&
            INNFER_FUSE_LAST && net->fused_tail && !any_map && net->band_rows == 0 && net->final_act == 0 && net->outm == 0) {
            // The last upconv_block -> HR_conv0 -> conv_last as ONE kernel (hr_chain.hip): the [64 ch, 2h x 2w] tensor between them (4.25 GB of a 1080p -> 4K frame, written
            // by a store-bound launch and read back by the next) stays in LDS tile by tile.  Same operands in the same order per value as the two launches below: same bits.
            const ConvSlot& ch = net->convs[ci];           // HR_conv0
            const ConvSlot& cl = net->convs[ci + 1];       // conv_last
            const int hh = 2 * h, wh = 2 * w;
            const long gh = (long)N * hh * wh * 32;
            if (ch.K == 64 && ch.C == 64 && ch.rowp && cl.d_fuse && cl.loaded && conv_fuse_side_bytes(N, hh, wh) <= (size_t)gh * 2 * (net->nf / 32)) {
                ConvLaunch L = mk(ch, nullptr, gh, ws + cv.hr, gh, N, hh, wh, net->trunk_act);
                L.fuse_w = (const f16*)cl.d_fuse; L.fuse_bias = cl.d_b; L.fuse_side = (float*)(ws + cv.hr); L.fuse_out = d_out; L.fuse_oc = cl.K;
                L.fuse_out_mode = out_dtype == INNFER_U8 ? 2 : out_dtype == INNFER_F32 ? 1 : 0;
                L.out_denorm = net->u8_normalize; L.out_round16 = net->u8_round16;
                if (hr_chain_ok(L)) {
                    rc = timed_begin(s);
                    if (rc) return rc;
                    {
                        thread_local unsigned parity = 0;
                        L.rev = INNFER_KNOB("INNFER_TILE_REV", 1) ? (int)(parity++ & 1) : 0;
                    }
                    rc = hr_chain_launch(L, t, gi, (const f16*)cs.d_up4p, cs.d_b4, net->trunk_act, s);
                    if (rc) return rc;
                    rc = debug_after("hr_chain", s);
                    if (rc) return rc;
                    // the three layers' ALGORITHMIC FLOPs (nine taps each on the HR grid); bytes: the LR tensor in, the planar result out, the weights
                    const double pxh = (double)N * hh * wh;
                    const double ob = L.fuse_out_mode == 2 ? 1.0 : L.fuse_out_mode == 1 ? 4.0 : 2.0;
                    return timed_end(s, 2.0 * 9.0 * (64.0 * 64 + 64.0 * 64 + 64.0 * cl.K) * pxh, pxh / 4 * 128.0 + pxh * cl.K * ob + 9.0 * (2 * 64.0 * 64 + 64.0 * cl.K) * 2.0, 5000 + 16 * 4);
                }
            }
        }
        if (!net->ps_up && net->up_phases && !net->fp32 && cs.d_up4 && !cs.d_map && net->trunk_act <= 2) {
            // Upsample(nearest 2x) -> conv -> act as the four output phases of the equivalent transposed conv: 2x2 taps on the LR grid instead of 3x3 on the
            // HR grid (2.25 x fewer MACs), on the phase-lattice instantiation (conv3x3_pc<.., TM = 0x1B>)
            ConvLaunch L = mk(cs, t, gi, dst, go, N, h, w, net->trunk_act);
            const bool one_visit = net->up_phases == 1 && cs.d_up4p && w > 16;       // all four phases in one visit of a tile (C = 64, plane-order panels); else one phase per visit
            L.wpk = (const f16*)(one_visit ? cs.d_up4p : cs.d_up4); L.bias = cs.d_b4;
            L.K = 4 * cs.K; L.phase_c = cs.K; L.deconv_phases = one_visit ? 1 : 2; L.rowp = one_visit ? 1 : 0;
            rc = do_conv(L, s);
        } else if (!net->ps_up) {    // Upsample(nearest 2x) -> conv -> act
            ConvLaunch L = mk(cs, t, gi, dst, go, N, 2 * h, 2 * w, net->trunk_act);
            L.up = 1;
            rc = do_conv(L, s);
        } else {                     // conv nf->4nf -> PixelShuffle(2) -> act (SRGAN: ReLU; RRDBNet(upsample_mode='pixelshuffle'): its act_type)
            ConvLaunch L = mk(cs, t, gi, dst, go, N, h, w, net->trunk_act);
            L.out_mode = OUT_SHUFFLE2;
            // the store of the producer / consumer kernel (conv3x3_pc PSH) while an output group stays below 2 GiB (conv_launch's bound for that form: chop batches of
            // more than 209 tiles of 200 x 200 at the first stage and 4x frames beyond 8.39 M LR pixels at the second take the two-workgroup kernel, whose indices are 64-bit)
            if (cs.d_ps && net->ps_pc && !cs.d_map && (long)N * h * w * 256 < 0x7fffffffL) { L.wpk = (const f16*)cs.d_ps; L.bias = cs.d_bps; L.rowp = 2; }
            rc = do_conv(L, s);
        }
        if (rc) return rc;
        t = dst; h *= 2; w *= 2;
    }
    {
        const ConvSlot& cs = net->convs[ci++];
        const long gh = (long)N * h * w * 32;
        ConvLaunch L = mk(cs, t, gh, ws + cv.hr, gh, N, h, w, net->trunk_act);
        // HR_conv0 -> conv_last in one kernel (conv3x3.hip, FUSE) where the shapes allow it: fp16 engine, 64 features, <= 3 planar float outputs without final
        // activation / outm, whole 16 x 32 tiles.  The rim buffer lives where the HR slab would have been.
        const ConvSlot& cl = net->convs[ci];
        if (INNFER_FUSE_LAST && net->fused_tail && !net->fp32 && !any_map && net->band_rows == 0 && cl.d_fuse && cl.loaded && net->final_act == 0 && net->outm == 0 &&
            (out_dtype == INNFER_F16 || out_dtype == INNFER_F32 || out_dtype == INNFER_U8) && conv_fuse_side_bytes(N, h, w) <= (size_t)gh * 2 * (net->nf / 32)) {
            L.fuse_w = (const f16*)cl.d_fuse; L.fuse_bias = cl.d_b; L.fuse_side = (float*)(ws + cv.hr); L.fuse_out = d_out; L.fuse_oc = cl.K;
            L.fuse_out_mode = out_dtype == INNFER_U8 ? 2 : out_dtype == INNFER_F32 ? 1 : 0;
            L.out_denorm = net->u8_normalize; L.out_round16 = net->u8_round16;     // (uint8: tensor2np as the store, as in the two-launch form)
            if (conv_fuse_last_ok(L)) {
                rc = do_conv(L, s);
                if (rc) return rc;
                return INNFER_OK;
            }
            L.fuse_w = nullptr;
        }
        rc = do_conv(L, s);
        if (rc) return rc;
    }
    {
        const ConvSlot& cs = net->convs[ci++];
        ConvLaunch L = mk(cs, (const f16*)(ws + cv.hr), (long)N * h * w * 32, d_out, 0, N, h, w, net->final_act);
        L.out_mode = OUT_NCHW; L.out_f32 = out_dtype == INNFER_F32;
        L.outm = net->outm;
        L.out_u8 = out_dtype == INNFER_U8; L.out_denorm = net->u8_normalize; L.out_round16 = net->u8_round16;     // tensor2np as the conv's epilogue
        rc = do_conv(L, s);
        if (rc) return rc;
    }
    return INNFER_OK;
}

extern "C" int innfer_net_forward_timed(innfer_net_t net, const void* d_in, int in_dtype, void* d_out, int out_dtype,
                                        int N, int H, int W, void* d_ws, size_t ws_bytes, void* stream,
                                        int cap, float* h_ms, double* h_flops, double* h_bytes, int* h_kind, int* n_launches) {
    LaunchTimer t;
    g_timer = &t;
    int rc = innfer_net_forward(net, d_in, in_dtype, d_out, out_dtype, N, H, W, d_ws, ws_bytes, stream);
    g_timer = nullptr;
    if (rc == INNFER_OK && hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
        rc = set_error(INNFER_ERR_HIP, "forward_timed: stream synchronize failed");
    const int n = (int)t.flops.size();
    if (rc == INNFER_OK) {
        if (n_launches) *n_launches = n;
        for (int i = 0; i < n && i < cap; ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, t.ev[2 * i], t.ev[2 * i + 1]);
            if (h_ms) h_ms[i] = ms;
            if (h_flops) h_flops[i] = t.flops[i];
            if (h_bytes) h_bytes[i] = t.bytes[i];
            if (h_kind) h_kind[i] = t.kind[i];
        }
    }
    for (auto e : t.ev) (void)hipEventDestroy(e);
    return rc;
}

// ---------------------------------------------------------------- single conv
extern "C" size_t innfer_conv3x3_packed_bytes(int K, int C) {
    if (K <= 0 || C <= 0 || C % 32) return 0;
    return conv_packed_bytes(K, C);
}

extern "C" int innfer_pack_conv3x3(const float* w, int K, int C, void* h_packed) {
    if (!w || !h_packed || K <= 0 || C <= 0 || C % 32)
        return set_error(INNFER_ERR_INVALID, "pack_conv3x3: K=%d C=%d (C must be a multiple of 32)", K, C);
    conv_pack(w, K, C, h_packed);
    return INNFER_OK;
}

extern "C" int innfer_pack_conv3x3_rows(const float* w, int K, int C, int plane_rows, void* h_packed) {
    if (!w || !h_packed || K <= 0 || C <= 0 || C % 32 || (plane_rows && K % 64))
        return set_error(INNFER_ERR_INVALID, "pack_conv3x3_rows: K=%d C=%d (C must be a multiple of 32; the plane row order needs K %% 64 == 0)", K, C);
    conv_pack(w, K, C, h_packed, plane_rows ? 1 : 0);
    return INNFER_OK;
}

extern "C" int innfer_pack_conv3x3_shuffle2(const float* w, const float* bias, int K, int C, void* h_packed, float* h_bias_out) {
    if (!w || !h_packed || !h_bias_out || K <= 0 || K % 256 || C <= 0 || C % 32)
        return set_error(INNFER_ERR_INVALID, "pack_conv3x3_shuffle2: K=%d C=%d (K must be a multiple of 256: four phases of 64-channel groups; C of 32)", K, C);
    conv_pack_shuffle2(w, bias, K, C, h_packed, h_bias_out);
    return INNFER_OK;
}

extern "C" int innfer_pack_convt2x_rows(const float* w, int K, int C, int k, int plane_rows, void* packed) {
    if (!w || !packed || K <= 0 || K % 64 || C <= 0 || C % 32 || (k != 3 && k != 4)) return set_error(INNFER_ERR_INVALID, "pack_convT2x_rows: K=%d (%% 64) C=%d (%% 32) k=%d (3 | 4)", K, C, k);
    conv_pack_deconv2x(w, K, C, k, packed, plane_rows ? 1 : 0);
    return INNFER_OK;
}

extern "C" int innfer_pack_conv3x3_split(const float* w, int K, int C, void* h_packed) {
    if (!w || !h_packed || K <= 0 || C <= 0 || C % 32)
        return set_error(INNFER_ERR_INVALID, "pack_conv3x3_split: K=%d C=%d (C must be a multiple of 32)", K, C);
    conv_pack_split(w, K, C, h_packed);
    return INNFER_OK;
}

extern "C" size_t innfer_conv7x1_packed_bytes(int K, int C) { return (K > 0 && C > 0 && C % 32 == 0) ? conv_packed_bytes7v(K, C) : 0; }
extern "C" int innfer_pack_conv7x1(const float* w, int K, int C, void* packed) {
    if (!w || !packed || K <= 0 || K % 32 || C <= 0 || C % 32) return set_error(INNFER_ERR_INVALID, "pack_conv7x1: K=%d (%% 32) C=%d (%% 32)", K, C);
    conv_pack7v(w, K, C, packed);
    return INNFER_OK;
}
extern "C" size_t innfer_conv4x4s2_packed_bytes(int K, int C) { return (K > 0 && C > 0 && C % 32 == 0) ? conv_packed_bytes_s2k4(K, C) : 0; }
extern "C" int innfer_pack_conv4x4s2(const float* w, int K, int C, void* packed) {
    if (!w || !packed || K <= 0 || K % 64 || C <= 0 || C % 32) return set_error(INNFER_ERR_INVALID, "pack_conv4x4s2: K=%d (%% 64) C=%d (%% 32)", K, C);
    conv_pack_s2k4(w, K, C, packed);
    return INNFER_OK;
}
extern "C" size_t innfer_convt2x_packed_bytes(int K, int C) { return (K > 0 && C > 0 && C % 32 == 0) ? conv_packed_bytes_deconv2x(K, C) : 0; }
extern "C" int innfer_pack_convt2x(const float* w, int K, int C, int k, void* packed) {
    if (!w || !packed || K <= 0 || K % 64 || C <= 0 || C % 32 || (k != 3 && k != 4)) return set_error(INNFER_ERR_INVALID, "pack_convT2x: K=%d (%% 64) C=%d (%% 32) k=%d (3 | 4)", K, C, k);
    conv_pack_deconv2x(w, K, C, k, packed);
    return INNFER_OK;
}

extern "C" int innfer_conv3x3_f16(const innfer_conv_args* a, void* stream) {
    if (!a || !a->d_in || !a->d_packed || !a->d_bias || !a->d_out) return set_error(INNFER_ERR_INVALID, "conv3x3: null argument");
    if (a->column7) {
        if (a->K <= 0 || a->K % 32 || a->K > 64 || a->out_ch_off || a->row_begin || a->row_end || a->dilation > 1 || a->dilation_groups || a->pixel_shuffle2 ||
            a->upsample2x || a->d_res1 || a->d_res2 || a->stride2_k4 || a->transposed2x || a->act < 0 || a->act > 2)
            return set_error(INNFER_ERR_UNSUPPORTED, "conv7x1: K %% 32 == 0, K <= 64 (K=%d), act 0..2, all rows, no residual / upsampling", a->K);
        ConvLaunch L{};
        L.in = (const f16*)a->d_in; L.in_gstride = a->in_group_stride; L.C = a->C;
        L.wpk = (const f16*)a->d_packed; L.bias = a->d_bias;
        L.out = a->d_out; L.out_gstride = a->out_group_stride; L.K = a->K;
        L.N = a->N; L.H = a->H; L.W = a->W; L.act = a->act; L.s1 = L.s2 = 1.f; L.y0 = 0; L.y1 = a->H; L.out_mode = OUT_SLAB;
        L.conv7v = 1; L.reflect = a->reflect_pad ? 1 : 0;
        return conv_launch(L, (hipStream_t)stream);
    }
    if (a->stride2_k4 || a->transposed2x) {
        if ((a->stride2_k4 && a->transposed2x) || (a->transposed2x && a->transposed2x != 3 && a->transposed2x != 4) || a->K <= 0 || a->K % 64 || a->out_ch_off ||
            a->row_begin || a->row_end || a->dilation > 1 || a->dilation_groups || a->reflect_pad || a->pixel_shuffle2 || a->upsample2x || a->d_res1 || a->d_res2 ||
            a->act < 0 || a->act > 2)
            return set_error(INNFER_ERR_UNSUPPORTED, "conv: the stride-2 forms need K %% 64 == 0 (K=%d), act 0..2, all rows, zero padding, no residual / upsampling", a->K);
        ConvLaunch L{};
        L.in = (const f16*)a->d_in; L.in_gstride = a->in_group_stride; L.C = a->C;
        L.wpk = (const f16*)a->d_packed; L.bias = a->d_bias;
        L.out = a->d_out; L.out_gstride = a->out_group_stride; L.K = a->K;
        L.N = a->N; L.H = a->H; L.W = a->W; L.act = a->act; L.s1 = L.s2 = 1.f; L.y0 = 0; L.y1 = a->H; L.out_mode = OUT_SLAB;
        if (a->stride2_k4) L.stride2 = 1;
        else { L.K = 4 * a->K; L.phase_c = a->K; L.deconv_phases = 1; L.rowp = a->plane_rows ? 1 : 0; }
        return conv_launch(L, (hipStream_t)stream);
    }
    if (a->pixel_shuffle2) {
        if (a->K <= 0 || a->K % 64 || a->out_ch_off || a->row_begin || a->row_end || a->dilation > 1 || a->dilation_groups || a->reflect_pad || a->act < 0 || a->act > 2)
            return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: the PixelShuffle(2) store needs K %% 64 == 0 (K=%d), out_ch_off 0, act 0..2, all rows, plain zero padding", a->K);
    } else if (a->dilation_groups > 0 ? a->K != 32 * a->dilation_groups : (a->K <= 0 || a->K % 16 || a->K > 64))
        return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: K=%d (need K %% 16 == 0, K <= 64; K = 32 * dilation_groups)", a->K);
    ConvLaunch L{};
    if (a->out_ch_off % 16 || (a->out_ch_off % 32 && a->K > 16))
        return set_error(INNFER_ERR_INVALID, "conv3x3: out_ch_off=%d must keep the %d output channels inside 32-channel groups", a->out_ch_off, a->K);
    L.in = (const f16*)a->d_in; L.in_gstride = a->in_group_stride; L.C = a->C;
    L.wpk = (const f16*)a->d_packed; L.bias = a->d_bias;
    L.out = (f16*)a->d_out + (long)(a->out_ch_off / 32) * a->out_group_stride; L.out_gstride = a->out_group_stride;
    L.out_coff = a->out_ch_off % 32; L.K = a->K;
    L.N = a->N; L.H = a->H; L.W = a->W; L.act = a->act; L.up = a->upsample2x;
    L.res1 = (const f16*)a->d_res1; L.res1_gstride = a->res1_group_stride; L.s1 = a->res1_scale;
    L.res2 = (const f16*)a->d_res2; L.res2_gstride = a->res2_group_stride; L.s2 = a->res2_scale;
    L.y0 = a->row_begin; L.y1 = a->row_end > 0 ? a->row_end : a->H;
    L.out_mode = a->pixel_shuffle2 ? OUT_SHUFFLE2 : OUT_SLAB; L.reflect = a->reflect_pad; L.dilation = a->dilation; L.dilation_groups = a->dilation_groups;
    L.res1_lds = a->res1_from_input ? 2 : 0;                 // (the single-conv call: wherever the shape qualifies, one residual or two)
    L.rowp = a->plane_rows ? 1 : 0;
    if (a->pixel_shuffle2 && a->plane_rows == 2) L.rowp = 2;          // phase-major panels + bias from innfer_pack_conv3x3_shuffle2: the producer / consumer kernel's store
    else if (a->pixel_shuffle2 && a->plane_rows) return set_error(INNFER_ERR_INVALID, "conv3x3: pixel_shuffle2 takes plane_rows 0 (innfer_pack_conv3x3 panels) or 2 (innfer_pack_conv3x3_shuffle2 panels)");
    if (a->reserved0) return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: innfer_conv_args.reserved0 (the row-Winograd experiment until ABI 111) must be 0");
    if (a->split) {
        if (a->pixel_shuffle2 || a->K % 32 || a->in_lo <= 0 || a->out_lo <= 0 || (a->d_res1 && a->res1_lo <= 0) || (a->d_res2 && a->res2_lo <= 0))
            return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3 (split): K in {32, 64} (K=%d), slab output, the lo distances of every tensor given", a->K);
        L.split = 1; L.in_lo = a->in_lo; L.out_lo = a->out_lo; L.res1_lo = a->res1_lo; L.res2_lo = a->res2_lo;
    }
    return conv_launch(L, (hipStream_t)stream);
}

extern "C" int innfer_nchw_to_slab(const void* d_src, int src_dtype, void* d_slab, int64_t group_stride, int ch_off,
                                   int N, int C, int H, int W, void* stream) {
    return nchw_to_slab(d_src, src_dtype == INNFER_F32, (f16*)d_slab, group_stride, ch_off, N, C, H, W, (hipStream_t)stream);
}

extern "C" int innfer_slab_to_nchw(const void* d_slab, int64_t group_stride, int ch_off, void* d_dst, int dst_dtype,
                                   int N, int C, int H, int W, void* stream) {
    return slab_to_nchw((const f16*)d_slab, group_stride, ch_off, d_dst, dst_dtype == INNFER_F32, N, C, H, W, (hipStream_t)stream);
}
