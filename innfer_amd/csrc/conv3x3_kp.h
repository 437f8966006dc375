// KP (the conv kernels' argument block) and the tile constants every kernel of the family shares.
// Part of csrc/conv3x3.hip, split out in round 6 so that csrc/hr_chain.hip (the up-conv -> HR_conv0 -> conv_last chain) can share the fused last conv
// (conv3x3_fuse.h) with it: no functional change.  Included inside namespace innfer { namespace { .. } }.  Not a stand-alone header.
constexpr int TW = 32;        // tile width (pixels)
constexpr int LWP = 36;       // LDS row pitch (pixels): 36 px = 2304 B = 9 x 256 B, so every row starts on bank 0
constexpr int LVALID = TW + 2;


struct KP {
    const f16* in; long in_img_stride; long in_gbytes; int nchunks;   // gbytes: bytes between channel groups
    const f16* wpk; const float* bias;
    void* out; long out_gstride; int out_coff;
    int K, KG;
    int H, W, Hs, Ws;
    int act, up;
    const f16* res1; long res1_gstride; float s1;
    const f16* res2; long res2_gstride; float s2;
    int y0, y1;
    int tiles_x, tiles_y;
    int out_f32;
    int outm;                // planar kernels: RRDBNet / SRResNet.forward(outm=...) after the activation
    int out_u8, out_denorm, out_round16;   // planar kernels with <= 4 channels: uint8 HWC BGR(A) image instead of planar floats (tensor2np as the epilogue)
    int N;
    int pf;                  // L2 prefetch of the next chunk's input lines
    int rev;                 // each XCD walks its run of tiles backwards
    int nrate, rate_start[9];// POLY kernels with nrate > 0: output channel group g (32 channels) is a conv of dilation g + 1 over its own tile grid;
                             // tiles [rate_start[g], rate_start[g+1]) of the launch belong to it (dil unused)
    int dil, fullH, fullW;   // POLY kernels: dilation d; H, W, N are those of the d*d polyphase sub-images (ceil(fullH/d) x ceil(fullW/d), N*d*d of them)
    int ncg;                 // S9 kernels: real 32-channel groups of the input (nchunks = 9 * ncg virtual chunks)
    float* stats_part;       // STATS kernels (TMF | 0x1000): per-(tile[, phase], consumer wave, channel) partial statistics (count, mean, M2) of the conv result
    int stats_cn;            //   channels of the output slab (K, or phase_c behind the phase lattice)
    int s9v;                 // S9 kernels: only the three VERTICAL displacements (a 7-tap column conv as three 3-tap blocks; nchunks = 3 * ncg)
    int reflect;             // out-of-image taps read the mirrored pixel (nn.ReflectionPad2d(1)) instead of zero; not with `up`
    int phase_c;             // OUT_NCHW: > 0 = channel ch is phase (ch / phase_c) of a 2x transposed conv: channel ch % phase_c at (2y + ph/2, 2x + ph%2)
    int total;               // tiles x channel groups of this launch
    int cv_gx, cv_gy, cv_h1, cv_w1;   // CV kernels (image canvas): the N images are the cells of a cv_gx x cv_gy grid, cell pitch (H + 1) x (W + 1)
    // FUSE kernels (TMF | 0x20000): the network's LAST conv (64 -> fl_oc <= 3 planar channels, no activation) inside this conv's epilogue -- see fused_last_epilogue
    const f16* fl_w;         //   its weights as four MFMA A fragments [row tile 2][k-step 2][lane 64][8] (conv_pack_fuse_last)
    const float* fl_bias;    //   fl_oc biases
    float* fl_side;          //   per tile 192 x 3 partial sums of the pixels within one pixel of a tile edge (conv_fuse_combine finishes them)
    void* fl_out;            //   the planar [N, fl_oc, H, W] result
    int fl_oc, fl_out_mode;  //   0 fp16, 1 fp32 planar; 2 the uint8 HWC image (out_denorm / out_round16 as for the planar kernels)
    const f16* sg_w; const float* sg_bias;   // SGATE kernels (TMF | 0x80000): a 1x1 conv of this conv's own fp16 result gates it -- out = v * sigmoid(W v + b): two A fragments [t][lane][8] (conv_pack_selfgate), 32 biases
    float rs1;               // RLDS kernels (TMF | 0x40000): 1 / s1 -- the residual res1 (= the conv's own input groups 0, 1) is added to the accumulators as x / s1 from the live LDS stage
    long in_lo_bytes;        // SPLIT kernels (TMF | 0x2000): the low-part twin of the input slab lies this many bytes behind it,
    long out_lo, res1_lo, res2_lo;   //   those of the output / residual slabs this many ELEMENTS behind them
#ifdef INNFER_ABLATE
    int abl;                 // diagnostic build only: 1 no stores, 2 no weight DMA, 4 no input DMA, 8 no MFMA phase
#endif
};

