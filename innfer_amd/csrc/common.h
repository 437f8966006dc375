// Shared declarations of the innfer_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../../include/innfer_amd.h"

namespace innfer {

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

int set_error(int code, const char* fmt, ...);

// Activations of the fp16 engines' epilogues on the hardware exponential / reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each): tanhf / expf / the IEEE division are 30 .. 50
// instructions per value, these 4 .. 6 -- the results are rounded to fp16 (or are a final fp32 image within the engines' 1e-2 bounds).  The fp32 mode (f32ops.hip) keeps libm.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }      // (+-inf -> +-1; absolute error ~1e-7)

// Generic launch timer (innfer_timer_start / innfer_timer_stop, net.hip): while a collection is open on the calling thread every instrumented launch is
// bracketed by a HIP-event pair on its stream and recorded with a kernel-family name and its ALGORITHMIC flops / bytes.  Off: two thread-local loads.
bool gt_on();
void gt_begin(hipStream_t s);
void gt_end(hipStream_t s, const char* name, double flops, double bytes);
struct GtScope {                                   // brackets the enclosing block
    hipStream_t s; const char* name; double flops, bytes; bool on;
    GtScope(hipStream_t s_, const char* n, double f, double b) : s(s_), name(n), flops(f), bytes(b), on(gt_on()) { if (on) gt_begin(s); }
    ~GtScope() { if (on) gt_end(s, name, flops, bytes); }
};

// A/B knobs of the kernel experiments kept under profiles/: an environment variable only in the diagnostic builds (`make ablate`,
// `make stamps`); the shipped library compiles the measured-best value in as a constant, so the untaken branches (and the kernel
// instantiations only they reach) are not even in the binary.  INNFER_DEBUG (net.hip) is the one run-time switch of the shipped library.
#ifdef INNFER_ABLATE
#define INNFER_KNOB(name, dflt) ([] { static const int v = getenv(name) ? atoi(getenv(name)) : (dflt); return v; }())
#else
#define INNFER_KNOB(name, dflt) (dflt)
#endif

#define INNFER_HIP(expr)                                                              \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess)                                                         \
            return innfer::set_error(INNFER_ERR_HIP, "%s failed: %s (%s:%d)", #expr,  \
                                     hipGetErrorString(_e), __FILE__, __LINE__);      \
    } while (0)

// ---- 3x3 convolution on fp16 NHWC slabs (conv3x3.hip) -----------------------
enum OutMode { OUT_SLAB = 0, OUT_NCHW = 1, OUT_SHUFFLE2 = 2 };

// Slab layout ("blocked NHWC"): channels in groups of 32; element (n,y,x,c) of a slab lives at
//   base + (c/32)*gstride + ((n*H + y)*W + x)*32 + c%32        (gstride in elements, >= N*H*W*32)
// so a 32-channel chunk of consecutive pixels is one contiguous run of full 128-B lines.
struct ConvLaunch {
    const f16* in; long in_gstride; int C;        // C % 32 == 0
    const f16* wpk; const float* bias;            // packed panels for KG*16*NT channels
    void* out; long out_gstride;                  // OUT_SLAB: pointer to the group holding channel 0 of the output
    int out_coff;                                 // channel offset inside that group (0 or 16)
    int K;                                        // valid output channels
    int N, H, W;                                  // conv (output) size
    int act; int up;                              // act: 0 none / 1 LeakyReLU(0.2) / 2 ReLU / 3 tanh (OUT_NCHW only) / 4, 5 res1 * sigmoid(conv) with / without LeakyReLU (slab) / 6 sigmoid (OUT_NCHW only) / 7 pair gate: 64 rows -> 32 outputs, row 16 lg + r (r < 8) times sigmoid(row 16 lg + r + 8) (slab, PAN's PAConv); up: input read through nearest 2x
    const f16* res1; long res1_gstride; float s1;
    const f16* res2; long res2_gstride; float s2;
    int y0, y1;                                   // output rows [y0,y1)
    int out_mode; int out_f32;                    // OUT_NCHW: planar, f16 or f32
    int rev;                                      // traverse the tiles in reverse order (speed only: see conv3x3.hip)
    int deconv_phases;                            // ConvTranspose2d(4, 2, 1) as four 2x2-tap phase convs: K = 4 * phase_c (phase-major, phase_c % 64 == 0), panels from
                                                  // conv_pack_taps(mask 0x1B), H x W = the INPUT grid, fp16 slab output of 2H x 2W pixels and phase_c channels
    int stride2;                                  // Conv2d(k 4, s 2, p 1) as the 2x2-tap conv of the space-to-depth source (gathered by the loader): H x W = the OUTPUT grid,
                                                  // the source slab holds 2H x 2W pixels; panels from conv_pack_taps(K, 4 * C, 0x1B0)
    float* stats_part;                            // != nullptr: the kernel also writes partial norm statistics of its result (count, mean, M2 per tile, consumer wave and
                                                  // channel; conv_stats_nper records per image, merged by norm::launch_combine_parts): 64-channel slab kernels, act 0, no residual
    int prefix_lrelu;                             // with conv1x1: the operand of input group k is LeakyReLU(0.2)(group 0 + .. + group k) (fp32 running sums): PPON's c2
    int conv1x1;                                  // 1x1 conv: the centre tap only is staged and multiplied (panels from conv_pack_1x1); slab outputs
    int dilation_groups;                          // G > 0: K = 32*G, output channel group g is the conv of dilation g+1 (own 32-output panel, panels of
                                                  // conv_pack(K=32) back to back, bias[32*G]): PPON's eight dilated convs in one launch
    int dilation;                                 // > 1: dilated 3x3 conv, zero padding = dilation (PPON); 32-output slab convs only
    int conv7v;                                   // 7 x 1 column conv, padding 3 rows (zero or `reflect`), panels from conv_pack7v: slab output, K % 32 == 0 -- the 7x7 first
                                                  // convs of the CycleGAN / WBC generators over their row-patch slab (channel kx * C + c holds the horizontally displaced input)
    int conv7;                                    // 7x7 conv, padding 3 (zero or `reflect`), panels from conv_pack7x7: OUT_NCHW, K <= 16 only
    int reflect;                                  // 1: ReflectionPad2d(1) instead of zero padding (slab / planar outputs of the producer-consumer kernel);
                                                  // 2: ReplicationPad2d(1) (3x3 slab convs)
    // HR_conv0 -> conv_last fused (conv3x3_pc<.., TMF | 0x20000>): this conv (64 -> 64, slab semantics, `out` unused) carries the network's last conv in its epilogue
    const f16* fuse_w;                            //   conv_pack_fuse_last() panel of the last conv (4 KB, device), nullptr = not fused
    const float* fuse_bias; float* fuse_side; void* fuse_out; int fuse_oc, fuse_out_mode;  // its bias, the rim buffer (conv_fuse_side_bytes), the result: 0 fp16 / 1 fp32 planar, 2 the uint8 HWC image (out_denorm, out_round16)
    int phase_c;                                  // OUT_NCHW: K = 4*phase_c channels are the 4 output phases of a stride-2 transposed conv (unet.hip)
    int outm;                                     // OUT_NCHW: `outm` of RRDBNet / SRResNet.forward applied after `act`: 1 (tanh + 1) / 2, 2 tanh, 3 sigmoid, 4 clamp(0, 1)
    int out_u8, out_denorm, out_round16;          // OUT_NCHW with <= 4 channels: write a uint8 HWC BGR(A) image instead -- tensor2np as the conv's epilogue
                                                  // (utils.py:197-248): [round to fp16,] optional denormalisation, clip(255 x).round() half to even, channel flip
    int split;                                    // fp32-accurate mode: every tensor is a (hi, lo) pair of fp16 slabs, lo = fp16((x - hi) * 2^11) (conv3x3.hip, SPLIT); panels from
    long in_lo, out_lo, res1_lo, res2_lo;         // conv_pack_split / conv_pack_1x1_split; *_lo: distance (elements) from a hi slab to its lo twin.  Plain 3x3 / 1x1 slab convs
                                                  // (act 0..2, residuals, upsampled input, batches) and the planar last conv
    int rowp;                                     // 2 (OUT_SHUFFLE2, K % 256 == 0): phase-major plane-order panels + bias from conv_pack_shuffle2 -- the PixelShuffle(2) store on the producer / consumer kernel.
                                                  // 1 (64-channel output groups, slab output): the panels have the PLANE row order (conv_pack*(.., rowp = 1)) -- a lane's sixteen channels
                                                  // are 16 bytes in each of the group's two slab planes, a store instruction touches one plane.  Plain 3x3 slab convs (residuals, canvas,
                                                  // RLDS, fused last conv) and the transposed-conv phases; not the split / statistics / gate / stride-2 / 1x1 / planar forms
    int in_relu;                                  // 1: the operand is max(stored input, 0) (applied as fragments leave LDS) -- transposed-conv phase launches and the planar
                                                  // <= 16-output kernel: the UNet stores a skip tensor once (LeakyReLU form) and its up conv reads relu(cat) from it
    const f16* gate_w; const float* gate_bias;    // != nullptr (32-output slab convs): out = act(v * sigmoid(W v + b)) with v = fp16(conv + bias) -- PAN's pixel attention behind an up-conv
                                                  // as this conv's epilogue; panels from conv_pack_selfgate (2 KB), 32 biases; `act` is the activation AFTER the gate
    int res1_lds;                                 // 1: when res1 is the conv's own input (groups 0, 1: the dense block's x5 * 0.2 + x), act 0 and K = 64, take it from the staged LDS tiles
                                                  // (conv3x3_pc RLDS: chunk order 2, 3, .., 0, 1; the residual enters the fp32 accumulators as x / s1) instead of re-reading it in the epilogue
};

// Panel geometry of packed weights.
int conv_nt_for(int K);                           // 16-channel tiles per group: 1, 2 or 4
size_t conv_packed_bytes(int K, int C);
void conv_pack_shuffle2(const float* w_oihw, const float* bias, int K, int C, void* packed, float* bias_out);   // host; phase-major plane-order panels of the PixelShuffle(2) store (ConvLaunch.rowp = 2, OUT_SHUFFLE2): K % 256 == 0
void conv_pack(const float* w_oihw, int K, int C, void* packed, int rowp = 0);   // host; rowp: the plane row order of 64-channel groups (ConvLaunch.rowp)
void conv_pack_split(const float* w_oihw, int K, int C, void* packed);     // host; 3 * conv_packed_bytes(K, C): the (wl | wh | wh) panels of ConvLaunch.split
void conv_pack_1x1_split(const float* w_oi, int K, int C, void* packed);   // host; 3 * conv_packed_bytes_taps(K, C, 0x10)
int conv_launch(const ConvLaunch& L, hipStream_t s);
size_t conv_packed_bytes_taps(int K, int C, int mask);
void conv_pack_selfgate(const float* w32x32, void* packed_2k);      // host; ConvLaunch.gate_w
void conv_pack_1x1(const float* w_oi, int K, int C, void* packed);    // host; w [K][C]
int conv_stats_nper(int H, int W, int phases);                              // host; see ConvLaunch.stats_part
size_t conv_packed_bytes7v(int K, int C);
void conv_pack7v(const float* w_oc7, int K, int C, void* packed);             // host; w [K][C][7]
size_t conv_packed_bytes_s2k4(int K, int C);
void conv_pack_s2k4(const float* w_oi44, int K, int C, void* packed);         // host; Conv2d(4, 2, 1) panels for ConvLaunch.stride2
size_t conv_packed_bytes_deconv2x(int K, int C);
bool conv_fuse_last_ok(const ConvLaunch& L);
size_t conv_fuse_side_bytes(int N, int H, int W);
void conv_pack_fuse_last(const float* w_last_oihw, int oc, void* packed_4k, int rowp = 0, int cin = 64);   // host; rowp = the row order of the HR_conv0 panel it is fused behind; cin = 32: behind a 32-channel conv (w [oc][32][3][3])
void conv_pack_up2x_phases(const float* w_oihw, int K, int C, void* packed, int rowp = 0);       // host; conv_packed_bytes_deconv2x(K, C) bytes: upconv_block as four 2x2-tap phases
void conv_pack_deconv2x(const float* w_io, int K, int C, int k, void* packed, int rowp = 0); // host; ConvTranspose2d(k = 3 | 4, 2, 1) panels for ConvLaunch.deconv_phases, w [C][K][k][k]
void conv_pack_taps(const float* w, int K, int C, int mask, void* packed, int rowp = 0);   // host; w [K][C][9], only the taps of `mask` are packed (conv_packed_bytes_taps)
size_t conv_packed_bytes7x7(int K, int C);
void conv_pack7x7(const float* w_oihw, int K, int C, void* packed);   // host; C % 32 == 0

// ---- the HR tail as one kernel (hr_chain.hip): the last upconv_block -> HR_conv0 -> conv_last chained through LDS ----
// L: the HR_conv0 launch carrying the fused last conv exactly as conv_launch would take it (fuse_* set; H x W = the HR grid; rowp 1); beside it the up-conv's input slab
// (two 32-channel groups on the H/2 x W/2 grid), its one-visit phase panels (conv_pack_up2x_phases(.., rowp 1)), the bias once per phase (256 floats) and its activation.
bool hr_chain_ok(const ConvLaunch& L);
int hr_chain_launch(const ConvLaunch& L, const f16* up_in, long up_in_gstride, const f16* up_wpk, const float* up_bias, int up_act, hipStream_t s);

// ---- first conv: few input channels, NCHW input (conv_first.hip) -------------
struct FirstConvLaunch {
    const void* in; int in_f32; int Cin;          // NCHW planar input
    int in_u8, in_norm, in_round16;               // in_u8: `in` is a uint8 HWC BGR(A) image instead -- np2tensor as the conv's prologue (utils.py:164-194):
                                                  // /255, channel flip, optional [-1,1] normalisation, optional rounding to fp16 (`.half()`, run.py:422)
    const float* w;                               // [Cin*9][K] fp32 (k-major), device
    const float* bias;
    f16* out; long out_gstride; f16* out2; long out2_gstride;
    int K; int N, H, W; int act;
    long out_lo, out2_lo;                         // != 0: split output (ConvLaunch.split): the lo part of every value goes this many elements behind its hi part
};
int first_conv_launch(const FirstConvLaunch& L, hipStream_t s);

// ---- layout / tiles / blend / pre-post (tiles.hip) ---------------------------
int nchw_to_slab(const void* src, int src_f32, f16* slab, long gstride, int ch_off, int N, int C, int H, int W, hipStream_t s);
int slab_to_nchw(const f16* slab, long gstride, int ch_off, void* dst, int dst_f32, int N, int C, int H, int W, hipStream_t s);

// ---- one SCPA block of PAN as one launch (pan_scpa.hip) -----------------------
size_t pan_scpa_blob_bytes();
// torch layouts, fp32: conv1_a / conv1_b [20][40], k1 / k3 / k4 [20][20][3][3], k2 [20][20] + bias [20], conv3 [40][40] -> the kernel's weight blob (host)
void pan_scpa_pack(const float* c1a, const float* c1b, const float* k1, const float* k2, const float* k2b, const float* k3, const float* k4, const float* c3, void* blob);
// in / out: slabs of two 32-channel groups (40 real channels, pad channels zero), group stride G elements; x -> x + conv3(cat[..]) (PAN_arch.py:58-105)
int pan_scpa_launch(const f16* in, f16* out, long G, const void* d_blob, int N, int H, int W, hipStream_t s, int in_c8 = 0, int out_c8 = 0, int duo = -1);      // duo: 1 two 4-wave workgroups per CU on 8 x 32 tiles, 0 one 8-wave workgroup on 16 x 32, < 0 by the frame      // *_c8: channels 32..39 as a compact 16-byte plane (between two SCPA blocks)

// ---- the same block in the fp32-accurate mode on (hi, lo) fp16 operand pairs (pan_scpa_split.hip) ----
// tensors as "split planes" of npx = N * H * W pixels: [hi ch 0..31: 64 B / pixel][hi 32..39: 16 B][lo 0..31: 64 B][lo 32..39: 16 B], 160 B per pixel in all
size_t pan_scpa_split_blob_bytes();
void pan_scpa_split_pack(const float* c1a, const float* c1b, const float* k1, const float* k2, const float* k2b, const float* k3, const float* k4, const float* c3, void* blob);
bool pan_scpa_split_ok(int N, int H, int W);                                         // 32-bit buffer offsets
int pan_scpa_split_launch(const void* in, void* out, const void* d_blob, int N, int H, int W, hipStream_t s);
int pan_split_from_nchw(const float* x, void* planes, int N, int H, int W, hipStream_t s);      // NCHW fp32, 40 channels
int pan_split_to_nchw(const void* planes, float* x, int N, int H, int W, hipStream_t s);

// ---- fp32 NCHW building blocks of the -no_fp16 mode of PAN / UNet (f32ops.hip) --------
struct F32Conv {
    const float* in; long in_nstride, in_cstride; int C, Hin, Win;       // input view: (n, c, y, x) at in + n * in_nstride + c * in_cstride + y * Win + x
    const float* wp; const float* bias; int K;                          // f32conv_pack panels; bias may be null
    float* out; long out_nstride, out_cstride, out_pstride; int Wout;    // output view: (n, k, Y, X) at out + n * ns + k * cs + (Y * Wout + X) * ps (ps 0 = 1)
    int Ho, Wo;                                                         // the grid this launch walks (the phase grid of a transposed conv)
    int osy, osx, ooy, oox;                                             // output pixel (Y, X) = (oy * osy + ooy, ox * osx + oox)
    int isy, isx;                                                       // tap source = (oy * isy + dy[tap], ox * isx + dx[tap]); outside the image: zero
    int ntap, dy[49], dx[49];                                           // up to 7 x 7 taps
    int pad_mode;                                                       // taps outside the image: 0 read zero, 1 the mirrored pixel (nn.ReflectionPad2d), 2 the border pixel (nn.ReplicationPad2d)
    int up;                                                             // taps walk the nearest-2x upsampled image (source pixel = coordinate >> 1)
    int in_act;                                                         // 0 / 1 LeakyReLU(0.2) / 2 ReLU applied to the input as it is read
    int act;                                                            // epilogue activation: 0 none, 1 LeakyReLU(0.2), 2 ReLU, 3 tanh, 4 sigmoid
    float oscale;                                                       // != 0: the activated value is multiplied by it before the residual is added (`x + 0.2 * conv`)
    const float* res; long res_nstride, res_cstride;                    // + residual after the activation (same pixel indexing as `out` with ps 1)
    const float* mul; long mul_nstride, mul_cstride;                    // v = mul * sigmoid(conv + bias) before the activation (pixel attention)
    int N;
    int phase_k;                                                        // > 0: the four output phases of a stride-2 transposed conv in ONE launch -- K = 4 * phase_k, channel k' is phase
                                                                        // k' / phase_k (a, b) = (ph >> 1, ph & 1) of output channel k' % phase_k at pixel (2 oy + a, 2 ox + b); bias / res / mul
                                                                        // are indexed by the output channel.  Pays where 4 * phase_k fits the 16-channel tile a single phase would take anyway
                                                                        // (the UNet's outermost layer, 3 channels): the input is read once instead of four times.  osy = osx = 1, oo* = 0.
};
size_t f32conv_packed_floats(int K, int C, int ntap);
void f32conv_pack(int K, int C, int ntap, const std::function<float(int, int, int)>& w, float* packed);     // host; w(k, c, tap)
int f32conv_launch(const F32Conv& L, hipStream_t s);
int f32_norm_launch(const float* in, long in_ns, long in_cs, float* out, long out_ns, long out_cs, int N, int C, long HW, int mode, float eps,
                    const float* weight, const float* bias, const float* rmean, const float* rvar, int act, hipStream_t s,
                    const float* res = nullptr, long res_ns = 0, long res_cs = 0);       // + res after the activation (a residual block's skip)
int f32_act_copy_launch(const float* in, long in_ns, float* out, long out_ns, long per_image, int N, int act, hipStream_t s);
int f32_maxpool4_launch(const float* in, float* out, long planes, int H, int W, hipStream_t s);
int f32_fsa_combine_launch(const float* att, int hp, int wp, int C, const float* inp, float* out, int N, int H, int W, const float* gamma, hipStream_t s,
                            float* att_t = nullptr);      // att_t: N * C * hp * wp floats of scratch -- the 4-pixel-strip form on channel-major rows (W == 4 wp)
int f32_upsample_launch(const float* in, float* out, long planes, int h, int w, int f, int bilinear, hipStream_t s);
int f32_upadd_launch(const float* in, const float* skip, float* out, long planes, int h, int w, int tf_mode, hipStream_t s);   // out = bilinear2x(in) + skip (WBCNet_arch.py:60-75; tf_mode: tf_2xupsample_bilinear :126-137)
int f32_axpy_launch(const float* x, const float* y, float* out, float a, long n, hipStream_t s);                 // out = a * x + y
int f32_prefix_lrelu_launch(float* t, int N, int groups, int gc, long hw, hipStream_t s);                           // channel group k <- LeakyReLU(0.2)(group 0 + .. + group k), in place (PPON_arch.py:104-114)

}  // namespace innfer
