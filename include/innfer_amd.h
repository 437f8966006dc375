/*
 * innfer_amd.h -- flat C ABI of the MI355X (gfx950) hot path for victorca25/iNNfer.
 *
 * Scope: generator forward (ESRGAN RRDBNet / SRResNet) +
 * chop_forward tile extraction and overlap blend + uint8<->float pre/post.
 * The reference has no FFI; each entry point below names the reference
 * interface (file:line in victorca25/iNNfer) it replaces.  INTEGRATION.md shows
 * the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 (INNFER_OK) or a negative innfer_status; no C++
 *     exception crosses the boundary; innfer_last_error() gives the thread's
 *     last message.
 *   - pointers named d_* are device (HBM) pointers owned by the caller (e.g.
 *     torch storages); h_* are host pointers.  `stream` is a hipStream_t passed
 *     as void* (NULL = default stream).  Nothing synchronises the stream.
 *   - activations inside the library are fp16 blocked-NHWC channel slabs, accumulated
 *     in fp32 on MFMA; tensors at the boundary are NCHW like the reference's
 *     torch tensors.
 *   - handles are immutable after the last innfer_net_set_conv(); forward is
 *     re-entrant across streams given distinct workspaces.
 */
#ifndef INNFER_AMD_H
#define INNFER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    INNFER_OK = 0,
    INNFER_ERR_INVALID = -1,     /* bad argument (the reference raises ValueError/AssertionError) */
    INNFER_ERR_HIP = -2,         /* HIP runtime error */
    INNFER_ERR_UNSUPPORTED = -3, /* valid for the reference, not built here (NotImplementedError) */
    INNFER_ERR_NOMEM = -4,
    INNFER_ERR_WORKSPACE = -5    /* workspace too small */
} innfer_status;

typedef enum { INNFER_F16 = 0, INNFER_F32 = 1, INNFER_U8 = 2 /* uint8 HWC image: innfer_net_forward only */ } innfer_dtype;

typedef struct innfer_net* innfer_net_t;

/* ABI revision of this header (major*100 + minor).  101/102: innfer_conv_args grew reflect_pad / dilation / dilation_groups (zero-initialise the struct),
 * innfer_wbc_create takes tf_mode, innfer_net_set_final_act.  103: innfer_net_forward_timed reports algorithmic bytes, innfer_conv_args.pixel_shuffle2, innfer_unet_set_eval,
 * innfer_comm_* / innfer_gather_tiles / innfer_shard_tiles.  104: innfer_rrdbnet_create_ex, innfer_pan_create_ex, innfer_srresnet_create_ex, innfer_resnet_create_ex, innfer_unet_create_ex, innfer_net_set_outm, innfer_guided_filter_ex, innfer_filter2d, innfer_net_set_pair_convs, innfer_inthwc_to_nchw / innfer_nchw_to_inthwc, innfer_linear_resize, INNFER_U8 at the network boundary (innfer_net_set_u8_io), innfer_extract_tiles_u8 / innfer_recompose_u8, innfer_conv_args.stride2_k4 / transposed2x / column7 with innfer_pack_conv4x4s2 / innfer_pack_convt2x / innfer_pack_conv7x1.  105: innfer_net_set_conv_input_map, SRResNet scale 3, PixelShuffle(3) stages (nf 64) and PixelShuffle(2) on nf 32.  106: the fp32-accurate mode -- innfer_net_set_precision, innfer_conv_args.split / *_lo, innfer_pack_conv3x3_split, innfer_nchw_to_slab_split / innfer_slab_split_to_nchw.  107: innfer_net_set_fused_tail, innfer_net_set_upconv_phases.  108: innfer_net_set_residual_lds, innfer_conv_args.res1_from_input, innfer_pan_set_fused_scpa, innfer_unet_set_precision, innfer_pan_set_precision, innfer_ppon_set_precision, innfer_resnet_set_precision, innfer_wbc_set_precision.  109: innfer_conv_args.plane_rows, innfer_pack_conv3x3_rows, innfer_pack_convt2x_rows; innfer_net_set_upconv_phases takes 0 / 1 / 2.  110: innfer_net_set_conv on a network in the fp32 mode builds that conv's split panels (either call order of set_precision / set_conv works); innfer_pack_conv3x3_shuffle2 + innfer_conv_args.plane_rows = 2.  111: innfer_net_set_hr_chain.  112: REMOVED -- innfer_net_set_pair_convs (csrc/conv_pair.hip: the fused conv pairs of a dense block, measured 3 % slower per frame in round 2 and off ever since), innfer_pack_conv3x3_wino / innfer_conv3x3_wino_packed_bytes and the meaning of innfer_conv_args.winograd (now reserved0, must be 0): the row-Winograd experiment of round 3.  113: no new symbol -- innfer_pan_set_precision(p, 1) now selects the split-operand forms for PAN's SCPA trunk / up-stages / attention (innfer_pan_set_fused_scpa(p, 0) keeps the 112 form; 5: A/B of the PA epilogue).  innfer_version() returns the library's; a binding should compare. */
#define INNFER_ABI_VERSION 113
int innfer_version(void);
const char* innfer_last_error(void);

/* ------------------------------------------------------------------ networks
 * Replaces architectures.get_network(opt_net) + nn.Module.forward
 * (architectures/__init__.py:5-40, RRDBNet_arch.py:16-62, SRResNet_arch.py:15-91).
 */

/* RRDBNet (old-arch ESRGAN).  gc is the dense growth (32 in the reference,
 * RRDBNet_arch.py:27); scale in {1,2,3,4,8,16} (3: one nearest-3x stage).  plus != 0 adds the ESRGAN+ paths
 * (x2 += conv1x1(x), x4 += x2: RRDBNet_arch.py:155-160; GaussianNoise is the identity in eval). */
int innfer_rrdbnet_create(innfer_net_t* out, int in_nc, int out_nc, int nf, int nb,
                          int gc, int scale, int plus);

/* The same with the graph-changing constructor arguments of RRDBNet_arch.py:16-48: nr = dense blocks per RRDB (3: parameters
 * `RDB1..RDB3`, else `RDBs.<i>`, RRDBNet_arch.py:73-88); act = `act_type` of every conv block (1 LeakyReLU(0.2), 2 ReLU);
 * pixelshuffle_up != 0 = upsample_mode 'pixelshuffle' (conv nf -> 4 nf, PixelShuffle(2), act: block.py:333-346; scale 3: conv nf -> 9 nf,
 * PixelShuffle(3), nf 64 only) instead of 'upconv'.  innfer_rrdbnet_create(...) = innfer_rrdbnet_create_ex(..., 3, 1, 0).  (104) */
int innfer_rrdbnet_create_ex(innfer_net_t* out, int in_nc, int out_nc, int nf, int nb,
                             int gc, int scale, int plus, int nr, int act, int pixelshuffle_up);

/* SRResNet / SRGAN with the reference defaults (norm none, ReLU, CNA,
 * pixelshuffle, res_scale 1: utils/defaults.py:53-67). */
int innfer_srresnet_create(innfer_net_t* out, int in_nc, int out_nc, int nf, int nb, int scale);
/* The same with the constructor arguments that keep the graph on the built kernels (SRResNet_arch.py:16-46,69-91): act = `act_type` (1 LeakyReLU(0.2),
 * 2 ReLU), res_scale (x + res * res_scale), upconv_up != 0 = upsample_mode 'upconv' (Upsample, conv, act) instead of 'pixelshuffle';
 * scale in {1,2,3,4,8} (3: one factor-3 stage; PixelShuffle(3) needs nf 64).  (104, 105) */
int innfer_srresnet_create_ex(innfer_net_t* out, int in_nc, int out_nc, int nf, int nb, int scale, int act, float res_scale, int upconv_up);
/* mode 'NAC' conv blocks (block.py:246-254: norm -> act -> conv; SRResNet's own default, SRResNet_arch.py:16-27): conv `idx` reads
 * act(alpha[c] * x + shift[c]) instead of x -- the eval-mode BatchNorm2d of its input channels as a per-channel map (NULL alpha / shift: 1 / 0) and
 * act (0 none, 1 LeakyReLU(0.2), 2 ReLU), one elementwise launch in front of the conv.  Built for the first conv of an SRResNet block and LR_conv
 * (the norm in front of a block's second conv follows the first conv and is folded into its weights by the host).  All of NULL, NULL, 0 removes the map.  (105) */
int innfer_net_set_conv_input_map(innfer_net_t net, int idx, const float* h_alpha, const float* h_shift, int act);

/* Arithmetic precision of innfer_net_forward, the reference's fp16 switch (`fp16 = not args.no_fp16 and gpu`, then `model.half()` / `t_img.half()`:
 * run.py:345,383,421-422).  fp32 = 0 (default): fp16 activations and weights, fp32 accumulation.  fp32 = 1: the fp32-accurate forward -- every activation
 * is kept as a PAIR of fp16 slabs (hi = fp16(x), lo = fp16((x - hi) * 2^11): 22 significant bits), every weight as a pair of panels, and a product is
 * xh wh + 2^-11 (xh wl + xl wh) on the fp16 matrix cores with fp32 accumulation; input fp32 or uint8, twice the workspace (ask innfer_net_workspace_bytes
 * after this call), three times the MFMA work.  Against the fp32 reference: <= 1e-4 on [0,1]-scaled outputs (measured ~1e-6).  Built for
 * RRDBNet / SRResNet (every constructor variant of innfer_*_create_ex); the other generators have their own fp32 mode (innfer_<api>_set_precision, 108).  (106)
 * A LOAD-TIME call: with fp32 = 1 it builds the split weight panels of every conv set so far (hipMalloc + synchronous copies; INNFER_ERR_NOMEM when they
 * do not fit -- the panels this call had built are released again) -- outside stream capture; innfer_net_forward itself never allocates.  Either call
 * order works: an innfer_net_set_conv on a network that is already in the fp32 mode builds that conv's panels itself (110). */
int innfer_net_set_precision(innfer_net_t net, int fp32);

void innfer_net_destroy(innfer_net_t net);

/* Convolutions in forward order.  key is the state-dict prefix in the
 * reference's naming ("model.1.sub.0.RDB1.conv1.0"); weights are
 * key+".weight" [K,C,3,3] and key+".bias" [K]. */
int innfer_net_num_convs(innfer_net_t net);
int innfer_net_conv_info(innfer_net_t net, int idx, char* key, size_t key_cap, int* K, int* C);

/* Replaces net.load_state_dict for one conv (run.py:93): host fp32 OIHW
 * weights + bias are packed into the MFMA panel layout and uploaded.
 * Synchronous (load time, not forward time). */
int innfer_net_set_conv(innfer_net_t net, int idx, const float* h_weight_oihw, const float* h_bias);

int innfer_net_scale(innfer_net_t net);
size_t innfer_net_workspace_bytes(innfer_net_t net, int N, int H, int W);

/* nn.Module.forward(x): d_in [N,in_nc,H,W] -> d_out [N,out_nc,s*H,s*W], NCHW,
 * dtypes per innfer_dtype.  Replaces `self.model(data)` (run.py:187-189,217-219). */
int innfer_net_forward(innfer_net_t net, const void* d_in, int in_dtype, void* d_out, int out_dtype,
                       int N, int H, int W, void* d_workspace, size_t workspace_bytes, void* stream);

/* Same forward with a HIP-event pair around every kernel launch (on `stream`), then a
 * stream synchronise.  Fills up to `cap` entries: elapsed ms, algorithmic FLOPs, algorithmic HBM
 * bytes (every operand read once, every result written once: (C + K) * 2 B per output pixel,
 * + K * 2 B per residual, + the weight panel) and kernel kind (0 = first conv; 16*NT +
 * out_mode = the conv instantiation with NT 16-channel output tiles; + 1000 its fp32-accurate form,
 * + 2000 the fused HR_conv0 + conv_last launch, + 3000 an up-conv as four 2x2-tap phases, whose FLOPs
 * are the algorithmic ones of the nine-tap layer it replaces: it executes 4/9 of them).  Used by
 * bench.py for the roofline object (both roofs per kernel). */
int innfer_net_forward_timed(innfer_net_t net, const void* d_in, int in_dtype, void* d_out, int out_dtype,
                             int N, int H, int W, void* d_workspace, size_t workspace_bytes, void* stream,
                             int cap, float* h_ms, double* h_flops, double* h_bytes, int* h_kind, int* n_launches);

/* Scheduling knob: 0 = one launch per layer over the whole frame; R>0 = skewed
 * row bands of R rows through the RRDB trunk (working set kept Infinity-Cache
 * resident; identical results). */
/* uint8 images at the network boundary (104): innfer_net_forward accepts INNFER_U8 as in_dtype and / or out_dtype.  The input then is N
 * uint8 HWC BGR(A) images back to back and np2tensor (utils/utils.py:164-194: /255, BGR->RGB, optional [-1,1] normalisation, `.half()` in fp16
 * mode) runs as the first conv's prologue; the output is N uint8 HWC BGR(A) images and tensor2np (utils/utils.py:197-248: optional
 * denormalisation, clip(255 x).round() half to even, RGB->BGR) runs as the last conv's epilogue.  Bit-identical to the separate passes
 * innfer_u8hwc_to_nchw / innfer_nchw_to_u8hwc around a float forward.  normalize: the `normalize` / `denormalize` flag of both; fp16_mode != 0:
 * values are rounded to fp16 where the reference's fp16 mode holds an fp16 tensor (default 1). */
int innfer_net_set_u8_io(innfer_net_t net, int normalize, int fp16_mode);

int innfer_net_set_band_rows(innfer_net_t net, int rows);

/* The residual of a dense block's last conv (`x5 * 0.2 + x`, RRDBNet_arch.py:161-165) is that conv's own input channels 0..63.  With the LDS form the
 * kernel stages those two channel groups LAST and adds x / 0.2 to the fp32 accumulators from the staged LDS tile; the epilogue scales by 0.2 -- x is not
 * read a second time (265 MB of a 1327 MB launch at 1080p).  on = 1 (default): the last dense block of every RRDB, whose epilogue also adds the RRDB's own
 * residual (RRDBNet_arch.py:98) -- the measured gain (-3.7 % on those launches; the one-residual launches lose 0.9 % with it); on = 2: every dense block;
 * on = 0: epilogue loads everywhere.  The sum is formed in another order than fma(acc, 0.2, x) (fp32 either way): results agree across the settings to the
 * last rounding of the fp16 output, not bit for bit.  fp16 engine, nf = 64; everything else ignores the switch.  (108) */
int innfer_net_set_residual_lds(innfer_net_t net, int on);

/* Scheduling knob: the last two convs of RRDBNet / SRResNet (HR_conv0 -> LeakyReLU -> conv_last, RRDBNet_arch.py:36-42) as ONE kernel -- the last conv
 * runs in the epilogue of HR_conv0 on the tile that kernel has just produced (csrc/conv3x3.hip, FUSE: no halo recompute; the pixels within one pixel
 * of a tile edge are finished by a small second pass), and the HR feature slab (4.25 GB for a 1080p -> 4K frame) is neither written nor read.
 * 1 (default) = fused wherever the shapes allow it: fp16 engine, 64 features, <= 3 output channels (planar fp16 / fp32 or the uint8 image), no final activation / outm, HR frame of whole
 * 16 x 32 tiles, no row bands; every other case runs the two launches.  0 = always two launches.  The fused form adds the last conv's 576 products of
 * an output in a different order (fp32 either way): results agree to the last rounding of the fp16 output, not bit for bit.  (107) */
int innfer_net_set_fused_tail(innfer_net_t net, int on);
/* on (default): the LAST upconv_block -> HR_conv0 -> conv_last (RRDBNet_arch.py:31-42, block.py:348-361) run as ONE kernel chained through LDS where the fused tail and the
 * one-visit up-conv both apply (fp16 engine, 64 features, whole 16 x 32 output tiles): the 64-channel HR tensor between them is neither written nor read.  Bit-identical to
 * the two launches it replaces (off).  ABI 111. */
int innfer_net_set_hr_chain(innfer_net_t net, int on);

/* Scheduling knob: the conv of an upconv_block (nn.Upsample(nearest 2x) -> conv 3x3 -> act, block.py:348-361) as the four 2x2-tap output phases of the
 * equivalent ConvTranspose2d(4, 2, 1) on the LR grid -- 2.25 x fewer multiply-adds than nine taps on the HR grid.  The taps that meet the same LR pixel are
 * summed in fp32 and rounded to fp16 ONCE (the nine-tap form rounds each of them): the same linear map with a rounding of the same size, not the same bits.
 * 1 (default) = phases for the fp16 engine where the conv has 64 n output channels -- 64 -> 64 convs on grids wider than 16 pixels compute all four phases in ONE visit of
 * a tile (the input tile staged once); 2 = one phase per visit of a tile everywhere (the same bits as 1: A/B and parity tests) (109); 0 = the nine-tap form through
 * the upsampling loader.  (107) */
int innfer_net_set_upconv_phases(innfer_net_t net, int mode);

/* `finalact` of the reference constructors (RRDBNet_arch.py:45-48: an activation module after the last conv):
 * 0 none (default), 1 LeakyReLU(0.2), 2 ReLU, 3 tanh, 6 sigmoid. */
int innfer_net_set_final_act(innfer_net_t net, int act);

/* `outm` of RRDBNet.forward / SRResNet.forward (RRDBNet_arch.py:50-62): a range limiter applied to the network's output, after `finalact`:
 * 0 none (default), 1 'scaltanh' (tanh(x) + 1) / 2, 2 'tanh', 3 'sigmoid', 4 'clamp' to [0, 1] -- in the last conv's epilogue.  (104) */
int innfer_net_set_outm(innfer_net_t net, int outm);

/* Algorithmic FLOPs of one forward (2*MAC of every conv). */
double innfer_net_flops(innfer_net_t net, int N, int H, int W);

/* ------------------------------------------------------------- pix2pix UNet
 * Replaces UnetGenerator(norm=batch, upsample_mode=deconv).forward (architectures/UNet_arch.py:11-161)
 * as run.py runs it for `-a p2p_256 / unet_256` (meval=False: BatchNorm uses the statistics of the
 * current image, run.py:299-303).  A batch is N independent batch-1 forwards (per-image statistics).
 * Parameters are addressed by their state-dict key ("model.model.1.model.2.weight" ...): conv /
 * conv-transpose weights in PyTorch layout, BatchNorm weight and bias, and the running statistics, which only
 * innfer_unet_set_eval(u, 1) forwards read.
 */
typedef struct innfer_unet* innfer_unet_t;
int innfer_unet_create(innfer_unet_t* out, int in_nc, int out_nc, int num_downs, int ngf);
/* The same with norm_type 'instance' (UNet_arch.py:38-41,101-104): nn.InstanceNorm2d layers (no parameters, no running statistics: always the
 * statistics of the image, train() and eval() alike) and a bias on every conv (it only matters where no norm follows: the outermost and the
 * innermost down conv, the outermost up conv).  upconv: upsample_mode 'upconv' (UNet_arch.py:119-122,131-134,143-146; block.py:348-361) -- every
 * ConvTranspose2d(4, 2, 1) is Upsample(nearest 2x) + Conv2d(3x3, zero padding), keys `<i>.1.weight`.
 * innfer_unet_create(...) = innfer_unet_create_ex(..., 0, 0).  (104) */
int innfer_unet_create_ex(innfer_unet_t* out, int in_nc, int out_nc, int num_downs, int ngf, int instance_norm, int upconv);
void innfer_unet_destroy(innfer_unet_t u);
int innfer_unet_num_params(innfer_unet_t u);
int innfer_unet_param_info(innfer_unet_t u, int idx, char* key, size_t key_cap, int* ndim, int* shape4);
int innfer_unet_set_param(innfer_unet_t u, int idx, const float* h_data);
/* nn.Module.eval() / .train() (run.py:96-99): eval_mode != 0 normalises with the running statistics ("...running_mean" / "...running_var",
 * set like any other parameter; unset = a fresh BatchNorm's 0 / 1) as ATen's eval-mode batch_norm does: alpha = weight / sqrt(running_var + eps),
 * y = x * alpha + (bias - running_mean * alpha).  Default 0: statistics of the current image (how run.py runs pix2pix, meval=False). */
int innfer_unet_set_eval(innfer_unet_t u, int eval_mode);
/* The reference's fp16 switch for this generator (`fp16 = not args.no_fp16 and gpu`, run.py:345,421-422).  fp32 = 0 (default): the fp16 engine.  fp32 = 1: every
 * conv, norm and activation in fp32 on NCHW fp32 tensors (csrc/f32ops.hip: v_mfma_f32_16x16x4_f32, fp32 statistics) -- <= 1e-4 of the fp32 reference;
 * innfer_unet_forward then takes and returns INNFER_F32 only, and innfer_unet_workspace_bytes answers for this mode.  A load-time call (packs fp32 panels:
 * hipMalloc + synchronous copies); call it after the last innfer_unet_set_param.  (108) */
int innfer_unet_set_precision(innfer_unet_t u, int fp32);
size_t innfer_unet_workspace_bytes(innfer_unet_t u, int N, int H, int W);
double innfer_unet_flops(innfer_unet_t u, int N, int H, int W);
/* d_in [N,in_nc,H,W] -> d_out [N,out_nc,H,W] (tanh range), NCHW f16/f32; H, W multiples of 2^num_downs. */
int innfer_unet_forward(innfer_unet_t u, const void* d_in, int in_dtype, void* d_out, int out_dtype,
                        int N, int H, int W, void* d_workspace, size_t workspace_bytes, void* stream);

/* --------------------------------------------------------------------- PAN
 * Replaces PAN.forward (architectures/PAN_arch.py:163-222) with its SCPA blocks (PAN_arch.py:47-106),
 * PAConv (PAN_arch.py:24-45), the FSA SelfAttentionBlock on a 4x max-pooled map (block.py:398-473,
 * bicubic re-expansion) and the nearest pixel-attention up-blocks (block.py pa_upconv_block), as
 * utils/defaults.py:78-89 configures them: nf 40, unf 24, nb 16, scale 1/2/3/4, self_attention=True,
 * double_scpa=False, ups_inter_mode='nearest'.  Parameters are addressed by state-dict key
 * ("SCPA_trunk.3.PACnv.k2.weight" ...), PyTorch layout, fp32 host data.
 */
typedef struct innfer_pan* innfer_pan_t;
int innfer_pan_create(innfer_pan_t* out, int in_nc, int out_nc, int nf, int unf, int nb, int scale);
/* The same with PAN's graph-changing constructor arguments (PAN_arch.py:115-141): self_attention = 0 drops the FSA block (fea + trunk goes
 * straight to the upsampler), double_scpa != 0 runs a second SCPA trunk + `trunk_conv2` behind the first, bilinear_up != 0 =
 * ups_inter_mode 'bilinear' (the up-blocks' B.Upsample(2, mode), align_corners False: PAN_arch.py:11-19, block.py:286-323) instead of 'nearest'.
 * innfer_pan_create(...) = innfer_pan_create_ex(..., 1, 0, 0).  (104) */
int innfer_pan_create_ex(innfer_pan_t* out, int in_nc, int out_nc, int nf, int unf, int nb, int scale, int self_attention, int double_scpa, int bilinear_up);
void innfer_pan_destroy(innfer_pan_t p);
int innfer_pan_num_params(innfer_pan_t p);
int innfer_pan_param_info(innfer_pan_t p, int idx, char* key, size_t key_cap, int* ndim, int* shape4);
int innfer_pan_set_param(innfer_pan_t p, int idx, const float* h_data);
size_t innfer_pan_workspace_bytes(innfer_pan_t p, int N, int H, int W);
/* Schedule of the SCPA trunk (PAN_arch.py:58-105).  on = 1 (default): a block is ONE launch -- conv1_a | conv1_b, k1, k3 * sigmoid(k2), k4, conv3 and the
 * residual on a 16 x 32 pixel tile with a 2-pixel halo, intermediates in LDS / registers, the block's weights resident in LDS (csrc/pan_scpa.hip); on = 0:
 * five launches of the halo-tile conv kernel per block (rounds 1-3).  Same fp16 roundings of every intermediate tensor; the fp32 sums of conv1 and conv3 are
 * formed per 32-channel k-step in both, so results agree to the last rounding of the fp16 tensors.  on != 0 also moves the FSA block's attention
 * (softmax(f^T g) h over the pooled pixels, block.py:398-473) from the VALU kernel to the matrix cores: scores from fp16 (hi, lo) pairs -- fp32-accurate --,
 * exact row maxima, exp and sums in fp32, p and h as fp16 operands of the P V product; on = 2 keeps the VALU attention behind the fused blocks; on = 3 (110) keeps the
 * trunk tensors on two-group slabs between the fused blocks (default since 110: channels 32..39 travel as a compact 16-byte plane there; same bits); on = 4 (110) runs the
 * last stage's HRconv and conv_last as two launches (default since 110 where the full-resolution grid is whole 16 x 32 tiles: conv_last in HRconv's epilogue, the 24-channel
 * full-resolution tensor neither written nor read; same values to the summation order of conv_last).  (108)
 * In the fp32 mode (innfer_pan_set_precision(p, 1); since 113): on != 0 runs the SCPA blocks (csrc/pan_scpa_split.hip), the up-stages and the attention on
 * (hi, lo) fp16 operand pairs of the matrix cores, on = 0 every conv on the generic fp32 kernel and the attention on the VALU kernel (the form of 108 - 112);
 * on = 5 keeps the PA block of an up-stage as its own 1x1 launch instead of the up-conv's epilogue (A/B).
 * fp16 (113): the block kernel has two forms with the same bits -- one 8-wave workgroup per CU on 16 x 32 tiles, two 4-wave workgroups per CU on 8 x 32 tiles (x loaded straight into
 * MFMA fragments; the default) -- on = 6 / 7 force the second / the first (A/B). */
int innfer_pan_set_fused_scpa(innfer_pan_t p, int on);
/* The reference's fp16 switch for this generator (run.py:345,421-422), as innfer_unet_set_precision: fp32 = 1 runs PAN.forward with fp32-accurate arithmetic
 * -- <= 1e-4 of the fp32 reference; fp32 tensors in and out; a load-time call.  (108: every conv on the generic fp32 kernel of csrc/f32ops.hip, the FSA
 * attention on the fp32 VALU kernel.  113: the SCPA trunk, nearest-2x up-stages and the attention as three-MFMA products of (hi, lo = (x - hi) * 2^11) fp16
 * pairs with fp32 accumulation -- the SR engine's SPLIT convention; conv_first, the trunk conv, the f | g | h projections, bilinear / 3x stages stay on f32ops.) */
int innfer_pan_set_precision(innfer_pan_t p, int fp32);
/* d_in [N,in_nc,H,W] -> d_out [N,out_nc,scale*H,scale*W], NCHW f16/f32; H, W >= 4. */
int innfer_pan_forward(innfer_pan_t p, const void* d_in, int in_dtype, void* d_out, int out_dtype,
                       int N, int H, int W, void* d_workspace, size_t workspace_bytes, void* stream);

/* -------------------------------------------------------------------- PPON
 * Replaces PPON.forward with RRBlock_32 / _ResBlock_32 (architectures/PPON_arch.py:12-129): first conv,
 * nb residual-in-residual blocks of eight dilated 3x3 convs (rates 1..8) each, the structure / perception
 * branches and the three reconstruction heads, as utils/defaults.py:68-77 configures them (nf 64, nb 24,
 * alpha 1).  Returns the three outputs of the reference (content, structure, perceptual); run.py keeps the
 * last one (run.py:191-192).  Parameters are addressed by state-dict key ("CFEM.1.sub.3.RB2.d5.weight" ...).
 */
typedef struct innfer_ppon* innfer_ppon_t;
int innfer_ppon_create(innfer_ppon_t* out, int in_nc, int out_nc, int nf, int nb, int scale, float alpha);
void innfer_ppon_destroy(innfer_ppon_t p);
int innfer_ppon_num_params(innfer_ppon_t p);
int innfer_ppon_param_info(innfer_ppon_t p, int idx, char* key, size_t key_cap, int* ndim, int* shape4);
int innfer_ppon_set_param(innfer_ppon_t p, int idx, const float* h_data);
size_t innfer_ppon_workspace_bytes(innfer_ppon_t p, int N, int H, int W);
/* The reference's fp16 switch for this generator (run.py:345,421-422), as innfer_unet_set_precision: fp32 = 1 runs PPON.forward in fp32 on NCHW fp32 tensors
 * (csrc/f32ops.hip: the dilated convs are entries of the generic conv's tap table) -- <= 1e-4 of the fp32 reference; fp32 tensors in and out.  (108) */
int innfer_ppon_set_precision(innfer_ppon_t p, int fp32);
/* d_in [N,in_nc,H,W] -> d_out_{c,s,p} [N,out_nc,scale*H,scale*W], NCHW f16/f32; d_out_c / d_out_s may be NULL. */
int innfer_ppon_forward(innfer_ppon_t p, const void* d_in, int in_dtype, void* d_out_c, void* d_out_s, void* d_out_p,
                        int out_dtype, int N, int H, int W, void* d_workspace, size_t workspace_bytes, void* stream);

/* -------------------------------------------------------- CycleGAN ResNet
 * Replaces ResnetGenerator(norm=instance, padding=reflect, upsample=deconv).forward with ResnetBlock
 * (architectures/ResNet_arch.py:19-151; `-a resnet_9blocks / cg_6 ...`, utils/defaults.py:124-140): reflection-padded
 * 7x7 and 3x3 convs, two stride-2 convs, n_blocks residual blocks, two ConvTranspose2d(3,2,1,1), tanh; InstanceNorm2d
 * without affine parameters, statistics of the instance.  H, W multiples of 4, >= 16; ngf 32 / 64 / 96 / 128 (the presets use 64).
 */
typedef struct innfer_resnet* innfer_resnet_t;
int innfer_resnet_create(innfer_resnet_t* out, int in_nc, int out_nc, int ngf, int n_blocks);
/* The same with ResnetBlock's constructor arguments (ResNet_arch.py:104-146): padding 0 reflect / 1 replicate / 2 zero (the pad layer in front of
 * the two 3x3 convs of every block; the parameter indices inside `conv_block` follow), use_dropout != 0 = an nn.Dropout(0.5) behind the first
 * conv + norm + ReLU (identity in eval mode -- how run.py runs these generators -- but it shifts the second conv's index); upconv != 0 =
 * upsample_mode 'upconv': the two ConvTranspose2d stages become Upsample(nearest 2x) + Conv2d(3x3) (block.py:348-361; parameters `model.<i>.1`);
 * batch_norm != 0 = norm_type 'batch' (the constructor's default, ResNet_arch.py:19,39-49): nn.BatchNorm2d behind every conv but the last (parameters
 * and buffers `model.<i+1>.{weight,bias,running_mean,running_var,num_batches_tracked}`) and no bias on those convs.  (104) */
int innfer_resnet_create_ex(innfer_resnet_t* out, int in_nc, int out_nc, int ngf, int n_blocks, int padding, int use_dropout, int upconv, int batch_norm);
/* BatchNorm mode of a batch_norm generator, as innfer_unet_set_eval: 0 (nn.Module.train()) the statistics of the image, != 0 (eval(), Model's default)
 * the running statistics.  No effect on instance-norm generators.  (104) */
int innfer_resnet_set_eval(innfer_resnet_t r, int eval_mode);
void innfer_resnet_destroy(innfer_resnet_t r);
int innfer_resnet_num_params(innfer_resnet_t r);
int innfer_resnet_param_info(innfer_resnet_t r, int idx, char* key, size_t key_cap, int* ndim, int* shape4);
int innfer_resnet_set_param(innfer_resnet_t r, int idx, const float* h_data);
size_t innfer_resnet_workspace_bytes(innfer_resnet_t r, int N, int H, int W);
/* The reference's fp16 switch for this generator (run.py:345,421-422), as innfer_unet_set_precision: fp32 = 1 runs the forward in fp32 on NCHW fp32 tensors
 * (csrc/f32ops.hip) -- <= 1e-4 of the fp32 reference; fp32 tensors in and out; a load-time call.  (108) */
int innfer_resnet_set_precision(innfer_resnet_t r, int fp32);
int innfer_resnet_forward(innfer_resnet_t r, const void* d_in, int in_dtype, void* d_out, int out_dtype,
                          int N, int H, int W, void* d_workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------- WBC UNet + guided filter
 * Replaces UnetGeneratorWBC(mode='pt' or 'tf').forward with ResBlock (architectures/WBCNet_arch.py:8-99; `-a wbcunet`) and
 * guided_filter(x, y, r=1, eps) in 'regular' mode (utils/utils.py:548-626; run.py:427-429 applies it with eps 5e-3 to
 * (input image, network output)).  H, W multiples of 4 (run.py modcrops to 4).
 */
typedef struct innfer_wbc* innfer_wbc_t;
int innfer_wbc_create(innfer_wbc_t* out, int nf, int tf_mode);   /* tf_mode: tf_same_padding + tf_2xupsample_bilinear (WBCNet_arch.py:126-142) */
void innfer_wbc_destroy(innfer_wbc_t u);
int innfer_wbc_num_params(innfer_wbc_t u);
int innfer_wbc_param_info(innfer_wbc_t u, int idx, char* key, size_t key_cap, int* ndim, int* shape4);
int innfer_wbc_set_param(innfer_wbc_t u, int idx, const float* h_data);
size_t innfer_wbc_workspace_bytes(innfer_wbc_t u, int N, int H, int W);
/* The reference's fp16 switch for this generator (run.py:345,421-422), as innfer_unet_set_precision: fp32 = 1 runs the forward in fp32 on NCHW fp32 tensors
 * (csrc/f32ops.hip) -- <= 1e-4 of the fp32 reference; fp32 tensors in and out; a load-time call.  (108) */
int innfer_wbc_set_precision(innfer_wbc_t w, int fp32);
int innfer_wbc_forward(innfer_wbc_t u, const void* d_in, int in_dtype, void* d_out, int out_dtype,
                       int N, int H, int W, void* d_workspace, size_t workspace_bytes, void* stream);
/* d_x (guidance), d_y (input), d_out: [N,C,H,W] tensors of one dtype (f16/f32); means over 3x3 windows, reflect padding. */
size_t innfer_guided_filter_workspace_bytes(int N, int C, int H, int W);
int innfer_guided_filter(const void* d_x, const void* d_y, int dtype, int N, int C, int H, int W, float eps, void* d_out,
                         void* d_workspace, size_t workspace_bytes, void* stream);
/* guided_filter with a ks x ks window (ks = 2 r + 1, odd) and, when d_x_hr is not null, its 'fast' mode (utils/utils.py:548-626): A and b of
 * the [N,C,H,W] pair are enlarged to [N,C,Hh,Wh] (bilinear, align_corners=True) and applied to the high-resolution guidance d_x_hr; d_out then is
 * [N,C,Hh,Wh].  Same workspace as innfer_guided_filter.  (104) */
int innfer_guided_filter_ex(const void* d_x, const void* d_y, int dtype, int N, int C, int H, int W, int ks, float eps,
                            const void* d_x_hr, int Hh, int Wh, void* d_out, void* d_workspace, size_t workspace_bytes, void* stream);

/* filter2D (utils/utils.py:484-535): every [H,W] plane of d_x (planes = B*C, f16 / f32) cross-correlated with one kH x kW fp32 kernel in device memory
 * behind F.pad(x, (pad_left, kW-1-pad_left, pad_top, kH-1-pad_top), mode); border 0 constant / 1 reflect / 2 replicate / 3 circular.  (104) */
int innfer_filter2d(const void* d_x, int dtype, long planes, int H, int W, const float* d_kernel, int kH, int kW, int pad_left, int pad_top,
                    int border, void* d_out, void* stream);

/* -------------------------------------------------- single fused convolution
 * The building block, exposed for tests: 3x3 stride-1 zero-pad-1 convolution
 * over an fp16 "blocked NHWC" channel slab (conv_block, block.py:213-254).
 * Slab layout: channels in groups of 32; element (n,y,x,c) lives at
 *   base + (c/32)*group_stride + ((n*H + y)*W + x)*32 + c%32   (elements; group_stride >= N*H*W*32)
 * so a 32-channel chunk of consecutive pixels is a contiguous run of full 128-byte lines and the
 * reference's torch.cat is a group offset.
 *   out[.., out_ch_off + k] = epilogue(conv(in[.., 0:C]))
 *   epilogue: +bias -> act -> (*res1_scale + res1) -> (*res2_scale + res2)
 * act: 0 none, 1 LeakyReLU(0.2) (block.py:89-90), 2 ReLU; 4 / 5: the pixel-attention gate of PAN (PAN_arch.py PA, PAConv):
 *   out = d_res1 * sigmoid(conv + bias), followed by LeakyReLU(0.2) for 4 (d_res1 required, no d_res2, scales unused);
 *   7: the pair gate of PAN's PAConv (PAN_arch.py:36-55: k3(x) * sigmoid(k2(x))) as ONE conv of K = 64 g rows that writes 32 g channels:
 *   row 16 q + r (r < 8) is value channel 8 q + r, row 16 q + 8 + r its gate -- out[8 q + r] = (conv + bias)[16 q + r] * sigmoid((conv + bias)[16 q + 8 + r])
 *   (no residuals; d_out holds K / 2 channels).
 * upsample2x: input is read through nearest-2x upsampling (block.py:321-322,358),
 *   i.e. d_in is [N,H/2,W/2,*] while H,W are the conv's (output) size.
 * d_packed comes from innfer_pack_conv3x3().  C % 32 == 0, K % 16 == 0, K <= 64 (pixel_shuffle2: K % 64 == 0, any K).
 */
typedef struct {
    const void* d_in;  int64_t in_group_stride;  /* elements between 32-channel groups of the input slab */
    int C;
    const void* d_packed; const float* d_bias;
    void* d_out; int64_t out_group_stride; int out_ch_off; int K;
    int N, H, W;
    int act; int upsample2x;
    const void* d_res1; int64_t res1_group_stride; float res1_scale;
    const void* d_res2; int64_t res2_group_stride; float res2_scale;
    int row_begin, row_end;             /* rows of the output to compute; 0,0 = all */
    int reflect_pad;                    /* != 0: nn.ReflectionPad2d(1) in front of the conv instead of zero padding (CycleGAN ResnetBlock,
                                           ResNet_arch.py:103-140); not together with upsample2x */
    int dilation;                       /* > 1: dilated 3x3 conv with zero padding = dilation (PPON's _ResBlock_32, PPON_arch.py:83-91); K == 32,
                                           no residuals / upsampling / row range */
    int dilation_groups;                /* G > 0 (<= 8): K = 32*G output channels, channel group g is the conv of dilation g+1 -- PPON's eight dilated
                                           convs as ONE launch; d_packed = the G panels of innfer_pack_conv3x3(K = 32) back to back, d_bias[32*G] */
    int pixel_shuffle2;                 /* != 0: nn.PixelShuffle(2) folded into the store (pixelshuffle_block, block.py:333-346): K % 64 == 0 conv channels,
                                           d_out is the [N,2H,2W,K/4] slab, out[n, c, 2y+i, 2x+j] = act(conv)[n, 4c+2i+j, y, x] -- a pure index map,
                                           bit-exact; act 0 / 1 / 2, residuals allowed, out_ch_off 0, no row range */
    int stride2_k4;                     /* != 0: nn.Conv2d(C, K, kernel_size=4, stride=2, padding=1) (the down convs of UnetSkipConnectionBlock, UNet_arch.py:105-106):
                                           d_in is the [N,2H,2W,C] slab, H x W the OUTPUT grid; d_packed from innfer_pack_conv4x4s2(); K % 64 == 0 (any K), act 0 / 1 / 2,
                                           no residuals / row range / out_ch_off (104) */
    int transposed2x;                   /* 3 | 4: nn.ConvTranspose2d(C, K, kernel_size=k, stride=2, padding=1, output_padding = 1 for k 3 / 0 for k 4) (UNet_arch.py:116-142,
                                           ResNet_arch.py:66-72): d_in is the [N,H,W,C] slab, d_out the [N,2H,2W,K] slab; d_packed from innfer_pack_convt2x(); d_bias holds
                                           the K biases FOUR times (one run per output phase); K % 64 == 0 (any K), act 0 / 1 / 2, no residuals / row range / out_ch_off (104) */
    int column7;                        /* != 0: nn.Conv2d(C, K, kernel_size=(7, 1), padding=(3, 0)) -- rows zero-padded, or reflected with reflect_pad (ReflectionPad2d rows):
                                           the 7x7 first convs of ResnetGenerator / UnetGeneratorWBC (ResNet_arch.py:57-60, WBCNet_arch.py:31) over their row-patch slab
                                           (channel kx * in_nc + c holds the horizontally displaced input); d_packed from innfer_pack_conv7x1(); K % 32 == 0, K <= 64,
                                           act 0 / 1 / 2, no residuals / row range / out_ch_off (104) */
    int split;                          /* != 0: the fp32-accurate form (innfer_net_set_precision): d_in / d_out / d_res1 / d_res2 are the HI slabs of (hi, lo) pairs whose
                                           lo slab = fp16((x - hi) * 2^11) lies in_lo / out_lo / res1_lo / res2_lo ELEMENTS behind; d_packed from innfer_pack_conv3x3_split();
                                           K in {32, 64} for slab outputs; act 0 / 1 / 2, residuals, upsample2x, row range and batches as for the fp16 form (106) */
    int64_t in_lo, out_lo, res1_lo, res2_lo;
    int reserved0;                      /* must be 0 (until ABI 111: `winograd`, the row-Winograd experiment of profiles/r3/winograd.txt -- measured slower, removed in 112; the
                                           field keeps the struct layout) */
    int res1_from_input;                /* != 0: d_res1 == d_in (same group stride), act 0, K = 64, C >= 96 -- the dense block's `x5 * 0.2 + x`: the residual is taken from the
                                           conv's own staged input tiles (see innfer_net_set_residual_lds); ignored when the shape does not qualify (108) */
    int plane_rows;                     /* 2 with pixel_shuffle2 (K % 256 == 0): d_packed / d_bias from innfer_pack_conv3x3_shuffle2() (110).
                                           != 0 (plain 3x3 slab convs and transposed2x with K % 64 == 0): d_packed comes from innfer_pack_conv3x3_rows(.., 1) /
                                           innfer_pack_convt2x_rows(.., 1) -- the row order in which a lane's sixteen output channels are 16 bytes in each of the group's two
                                           32-channel slab planes (what the networks use for their 64-output layers: one plane per store instruction); results are the same (109) */
} innfer_conv_args;

size_t innfer_conv3x3_packed_bytes(int K, int C);
int innfer_pack_conv3x3(const float* h_weight_oihw, int K, int C, void* h_packed);
int innfer_pack_conv3x3_rows(const float* h_weight_oihw, int K, int C, int plane_rows, void* h_packed);      /* (109) same size; plane_rows: see innfer_conv_args */
int innfer_pack_convt2x_rows(const float* h_weight_iohw, int K, int C, int k, int plane_rows, void* h_packed); /* (109) */
/* (110) Panels for pixel_shuffle2 with plane_rows = 2: the conv channels PHASE-MAJOR (packed channel ph * K / 4 + oc = reference channel 4 oc + ph, ph = 2a + b the position
 * in nn.PixelShuffle's 2 x 2 block) in the plane row order, so that the shuffle is the store of the producer / consumer kernel (what the networks use for their PixelShuffle(2)
 * stages on 64 features).  K % 256 == 0; same size as innfer_pack_conv3x3; h_bias_out receives the K biases in the same order (zeros when h_bias is NULL): pass it as d_bias. */
int innfer_pack_conv3x3_shuffle2(const float* h_weight_oihw, const float* h_bias, int K, int C, void* h_packed, float* h_bias_out);
/* Panels of the two stride-2 forms above: h_weight is torch's layout, [K][C][4][4] for the conv and [C][K][k][k] for the transposed conv (k = 3 | 4). (104) */
size_t innfer_conv7x1_packed_bytes(int K, int C);
int innfer_pack_conv7x1(const float* h_weight_oc7, int K, int C, void* h_packed);     /* h_weight [K][C][7] */
size_t innfer_conv4x4s2_packed_bytes(int K, int C);
int innfer_pack_conv4x4s2(const float* h_weight_oihw, int K, int C, void* h_packed);
size_t innfer_convt2x_packed_bytes(int K, int C);
int innfer_pack_convt2x(const float* h_weight_iohw, int K, int C, int k, void* h_packed);
int innfer_conv3x3_f16(const innfer_conv_args* a, void* stream);
/* Panels of the split form: 3 * innfer_conv3x3_packed_bytes(K, C) bytes ((w - wh) * 2^11 | wh | wh, in the order the kernel's virtual chunks meet them).  (106) */
int innfer_pack_conv3x3_split(const float* h_weight_oihw, int K, int C, void* h_packed);

/* NCHW (f16/f32) <-> blocked-NHWC f16 slab helpers used by tests of the single conv. */
int innfer_nchw_to_slab(const void* d_src, int src_dtype, void* d_slab, int64_t group_stride, int ch_off,
                        int N, int C, int H, int W, void* stream);
int innfer_slab_to_nchw(const void* d_slab, int64_t group_stride, int ch_off, void* d_dst, int dst_dtype,
                        int N, int C, int H, int W, void* stream);
/* The same for the (hi, lo) slab pairs of the fp32-accurate mode: fp32 NCHW -> hi slab at d_slab, lo slab `lo` elements behind it, and back
 * (hi + lo * 2^-11, exact in fp32).  (106) */
int innfer_nchw_to_slab_split(const float* d_src, void* d_slab, int64_t group_stride, int64_t lo, int ch_off, int N, int C, int H, int W, void* stream);
int innfer_slab_split_to_nchw(const void* d_slab, int64_t group_stride, int64_t lo, int ch_off, float* d_dst, int N, int C, int H, int W, void* stream);

/* ------------------------------------------------------------ launch timer (measurement only)
 * Between innfer_timer_start() and innfer_timer_stop() every kernel launch the calling thread makes through the library's conv / gather-GEMM / UNet
 * entry points is bracketed by a HIP-event pair on its stream.  innfer_timer_stop synchronises `stream` and returns, per launch in issue order, a
 * kernel-family name (names[i * name_cap ..], NUL-terminated), the elapsed ms, and the launch's ALGORITHMIC flops and HBM bytes (operands read once,
 * results written once; 0 where a family has none).  n receives the number of launches recorded (entries beyond cap are dropped).  bench.py builds
 * the per-kernel two-roof entries of its `unet64` object from it.  (106) */
int innfer_timer_start(void);
int innfer_timer_stop(void* stream, int cap, char* names, int name_cap, float* h_ms, double* h_flops, double* h_bytes, int* n);

/* ------------------------------------------------------------ chop / blend
 * Replaces extract_patches_2d / recompose_tensor (utils/utils.py:318-369,
 * 372-445) as used by Model.chop_forward (run.py:167-202).
 */

/* Geometry only (host).  ps = min(H,W,patch); origins are the row-major tile
 * lattice of stride int(ps*step) plus one ragged row/column anchored at H-ps /
 * W-ps.  ys/xs may be NULL to query counts.  Capacity: nh<=H, nw<=W. */
int innfer_chop_plan(int H, int W, int patch, double step, int* ps, int* nh, int* nw,
                     int* ys, int* xs);

/* d_img [B,C,H,W] -> d_tiles [nh*nw*?..]: tiles[(b? see below)]
 * Output order matches extract_patches_2d(batch_first=True).squeeze(0) for B=1:
 * d_tiles [n, C, ps, ps], n = nh*nw row-major.  tile_begin/tile_count select a
 * contiguous sub-range (multi-GPU sharding). */
int innfer_extract_tiles(const void* d_img, int dtype, int C, int H, int W, int patch, double step,
                         int tile_begin, int tile_count, void* d_tiles, void* stream);

/* 1-D blending profile (utils.py:396,413-416), host, fp32, length P. */
int innfer_blend_profile(int P, double step, int scale, float* h_profile);

/* recompose_tensor: d_tiles [n,C,P,P] (HR tiles, row-major over the tile
 * lattice) -> d_out [1,C,scale*height,scale*width].  Accumulates in fp32 in
 * the reference's (h,w) order: with fp32 tiles the result is bit-identical to
 * the reference's fp32 path. */
int innfer_recompose(const void* d_tiles, int dtype, int n, int C, int P, int height, int width,
                     double step, int scale, void* d_out, int out_dtype, void* stream);

/* ------------------------------------------------------------ multi-GPU chop (RCCL over xGMI)
 * The reference is single-process; what shards is chop_forward's tile list (run.py:186-197: every tile's forward is independent) and the
 * one exchange is "raw HR tiles -> rank 0" in front of recompose_tensor (utils/utils.py:372-445), plus the broadcast of the blended
 * intermediate between the models of a chain (run.py:424-426).  One process per GPU; the calls below are collective over the ranks of a
 * communicator.  RCCL (librccl.so.1) is bound at run time, on the first of these calls: single-GPU users never load it.
 */
#define INNFER_COMM_ID_BYTES 128
typedef struct innfer_comm* innfer_comm_t;

/* Host geometry: rank's contiguous share [first, first+count) of the row-major tile list, split as evenly as possible (earlier ranks take
 * the remainder: 798 tiles over 8 ranks -> 100 x 6, 99 x 2).  Use it with innfer_extract_tiles(tile_begin, tile_count). */
int innfer_shard_tiles(int n_tiles, int nranks, int rank, int* first, int* count);

/* ncclGetUniqueId on ONE rank; the caller ships the INNFER_COMM_ID_BYTES to the other ranks by its own means (file, socket, MPI, torch store). */
int innfer_comm_unique_id(void* h_id);
/* ncclCommInitRank on the calling thread's current HIP device. */
int innfer_comm_init(innfer_comm_t* out, const void* h_id, int rank, int nranks);
void innfer_comm_destroy(innfer_comm_t c);
int innfer_comm_rank(innfer_comm_t c);
int innfer_comm_size(innfer_comm_t c);

/* Collect every rank's tiles on rank 0, real tiles only.  Rank 0: d_tiles is the whole [n_tiles][tile_bytes] buffer the blend reads, with
 * its own share already in place; rank r > 0: d_tiles is its own share [count][tile_bytes] (innfer_shard_tiles).  One group of ncclSend /
 * ncclRecv enqueued on `stream`; the buffer may be read by work enqueued on `stream` after the call. */
int innfer_gather_tiles(innfer_comm_t c, void* d_tiles, size_t tile_bytes, int n_tiles, void* stream);
/* ncclBroadcast of `bytes` bytes from `root` (the blended intermediate of a model chain), in place. */
int innfer_comm_broadcast(innfer_comm_t c, void* d_buf, size_t bytes, int root, void* stream);

/* --------------------------------------------------------------- pre / post
 * np2tensor / tensor2np (utils/utils.py:164-194,197-248, colors.py:5-26):
 * uint8 HWC BGR(A) <-> float NCHW RGB(A), /255, optional [-1,1] (de)normalise,
 * clip*255 and round-half-to-even on the way back.
 */
int innfer_u8hwc_to_nchw(const uint8_t* d_img, int H, int W, int C, int normalize,
                         void* d_out, int out_dtype, void* stream);
int innfer_nchw_to_u8hwc(const void* d_in, int in_dtype, int H, int W, int C, int denormalize,
                         uint8_t* d_img, void* stream);
/* The same with the remaining arguments of np2tensor / tensor2np (104).  bits: 8 or 16 (uint8 / uint16 image, MAX_VALUES_BY_DTYPE
 * utils.py:22-33); maxval: what the image is divided by (255, 65535; 1 = change_range False); bgr2rgb / rgb2bgr = 0 keeps the channel order.
 * innfer_nchw_to_inthwc scales by the data_range that goes with the type (255 / 65535), clips and rounds half to even. */
/* The chop path with uint8 images at both ends (104): extract_patches_2d(np2tensor(img)) and tensor2np(recompose_tensor(tiles)) without the
 * float image in between -- the tile gather converts (same /255, flip, normalisation, cast to the tile dtype), the blend stores uint8 HWC
 * BGR(A) (the blended value is first rounded to via_dtype, the dtype recompose_tensor would have returned).  One image; same geometry and
 * arithmetic as innfer_extract_tiles / innfer_recompose, bit-identical to the separate passes. */
int innfer_extract_tiles_u8(const uint8_t* d_img, int C, int H, int W, int normalize, int patch, double step,
                            int tile_begin, int tile_count, void* d_tiles, int tile_dtype, void* stream);
int innfer_recompose_u8(const void* d_tiles, int dtype, int n_tiles, int C, int P, int height, int width, double step, int scale,
                        int via_dtype, int denormalize, uint8_t* d_img, void* stream);
int innfer_inthwc_to_nchw(const void* d_img, int bits, int H, int W, int C, int bgr2rgb, int normalize, float maxval,
                          void* d_out, int out_dtype, void* stream);
int innfer_nchw_to_inthwc(const void* d_in, int in_dtype, int H, int W, int C, int rgb2bgr, int denormalize, int bits,
                          void* d_img, void* stream);

/* srgb2linear / linear2srgb (utils/colors.py:29-46, 49-60), the pointwise halves of the `-cf` colour
 * fix: uint8 sRGB -> float32 linear, and float32 linear -> uint8 sRGB (clip, gamma, *255, TRUNCATING
 * cast like astype(np.uint8)).  n = number of elements (any layout). */
int innfer_srgb_to_linear(const uint8_t* d_in, float* d_out, size_t n, void* stream);
int innfer_linear_to_srgb(const float* d_in, uint8_t* d_out, size_t n, void* stream);

/* Colour fix (`-cf`): replaces color_fix (utils/utils.py:278-315).  d_lr / d_sr / d_out are uint8 HWC images
 * on the device (any channel order, C <= 4); out = linear2srgb(resize(gauss3(lin(lr) - resize(lin(sr)))) + lin(sr)).
 * The reference's cv2.resize(INTER_CUBIC) and cv2.GaussianBlur(3x3, sigma 0) are restated from OpenCV's
 * published float32 algorithms (parity with OpenCV itself is unpinned: it is absent from the image). */
size_t innfer_color_fix_workspace_bytes(int h_lr, int w_lr, int h_sr, int w_sr, int channels);
int innfer_color_fix(const uint8_t* d_lr, int h_lr, int w_lr, const uint8_t* d_sr, int h_sr, int w_sr, int channels,
                     uint8_t* d_out, void* d_workspace, size_t workspace_bytes, void* stream);

/* linear_resize (utils/utils.py:267-276, the pix2pix pre-step of run.py:413-414): uint8 HWC image -> srgb2linear -> bicubic resize to
 * (oh, ow) (OpenCV INTER_CUBIC, a = -0.75, as in innfer_color_fix: parity with OpenCV itself unpinned) -> linear2srgb, uint8 HWC.
 * Workspace: h * w * C floats.  (104) */
int innfer_linear_resize(const uint8_t* d_img, int h, int w, int C, uint8_t* d_out, int oh, int ow,
                         void* d_workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* INNFER_AMD_H */
