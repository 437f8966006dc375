// 3x3 stride-1 zero-pad-1 convolution as implicit GEMM on CDNA4 MFMA (gfx950).
//
// Replaces the reference's conv_block -> nn.Conv2d [+ LeakyReLU/ReLU] [+ torch.cat]
// [+ x*0.2 + residual] [+ nearest-2x Upsample in front] [+ PixelShuffle behind]
// (architectures/block.py:213-254,333-361; RRDBNet_arch.py:152-165,91-98).  The other generators reuse it for every
// conv it can express -- PAN (all convs; a 1x1 conv is a panel with a centre tap only; pixel-attention gate epilogue
// res1 * sigmoid(conv)), PPON, the WBC UNet, and the pix2pix UNet's outermost ConvTranspose (the four output phases as
// 4*out_nc channels of one 3x3 conv, tanh + phase scatter in the planar epilogue), the CycleGAN residual blocks (reflection padding in
// the loader), the 7x7 convs with few outputs of the WBC UNet / CycleGAN (conv3x3_pc<..,S9>: nine displaced 3x3 convs) and PPON's dilated
// convs (conv3x3_pc<..,POLY>: ordinary 3x3 convs on the polyphase components of the image).
//
// Data layout in HBM: activations are fp16 "blocked NHWC" channel slabs: channels in groups of
// 32, element (n,y,x,c) at base + (c/32)*group_stride + ((n*H+y)*W+x)*32 + c%32, so a 32-channel
// chunk of consecutive pixels is a contiguous run of full 128-B lines and the dense concat of the
// reference (x|x1|x2|x3|x4) is a group offset, never a copy.
//
// GEMM view per tap (r,s) and 32-channel chunk:  D[oc][px] += W[oc][c] * X[px+tap][c]
//   MFMA v_mfma_f32_16x16x32_f16, A = weights (rows = 16 out channels),
//   B = pixels (cols = 16 consecutive pixels of one image row), fp32 accumulate.
//   Output channels are permuted inside a panel so that one lane ends up holding
//   4*NT CONSECUTIVE channels of one pixel -> 8/16/32-byte vector stores into NHWC.
//
// Two kernels share the tile machinery below:
//   conv3x3_pc<RPW,NT,NLW,OUT>   producer / consumer workgroup (8 MFMA waves + NLW LDS-DMA waves, one per CU,
//                                tile = 8*RPW rows x 32 px, two LDS stages): every slab-output conv and the
//                                planar last conv; two more template flags change only the ADDRESSING of its loader /
//                                epilogue: S9 (a 7x7 conv as nine displaced 3x3 convs) and POLY (a dilated conv as
//                                ordinary convs on the polyphase components of the image);
//   conv3x3_mfma<RPW,NT,OUT>     two independent 4-wave workgroups per CU, tile = 4*RPW rows x 32 px, one LDS
//                                stage each (round 1's kernel; kept as the FALLBACK some launches still reach): PixelShuffle(2) stores whose output group is
//                                2 GiB or more (64-bit indices: SRResNet chop batches beyond 209 tiles, net.hip), PixelShuffle on plain panels (ABI plane_rows 0), planar
//                                outputs with residuals or 32 / 64 planar channels, and every slab conv of the diagnostic builds' INNFER_PC=0.
// Per 32-channel chunk the halo tile ((rows+2) x 34 px x 64 B) and the weight panel (9 x 16*NT x 64 B) are
// staged with LDS-DMA (buffer_load_dwordx4 ... lds; out-of-image lanes are zero-filled by the buffer range
// check = the conv's zero padding).  LDS rows are 36 px (2304 B = 9 x 256 B: every row starts on bank 0);
// the 16-B-slot XOR swizzle (slot ^= 2*bit2(pixel index)) then depends on (lane, tap column, row parity)
// only: all ds_read_b128 are base+immediate and bank-conflict free.  A pixel fragment B(row, seg, s) is
// read once and used by the up to three (row-in-wave, r) pairs that need it.
#include "common.h"
#include <type_traits>

#include <cstdlib>
#include <cstring>
#include <vector>

// Pins a value as a rounded fp32 number in a VGPR: the backend cannot fold the fp16 conversion that
// follows into the fma that produced it (v_fma_mixlo_f16 rounds ONCE, fma + convert rounds twice).
#define FP32_VALUE(x) asm("" : "+v"(x))
// cache-policy bits of the halo-tile LDS-DMA pieces (A/B builds only: -DINNFER_IN_AUX=2 = nt, 1 = sc0, 17 = sc0 sc1 ...); measured: profiles/r3/in_aux_ab.txt
#ifndef INNFER_IN_AUX
#define INNFER_IN_AUX 0
#endif

// No implicit floating-point contraction in this file.  With contraction allowed the backend may fold
// "round to fp16 of (a*b + c)" into v_fma_mixlo_f16 (ONE rounding) for some unrolled copies of the
// epilogue and keep fma + convert (TWO roundings) for others, which makes a pixel's value depend on
// where in a tile it was computed (1 fp16 ulp, seen as banded != plain schedule).  The residual adds
// below are explicit fmaf() calls, pinned with FP32_VALUE before the conversion.
#pragma clang fp contract(off)

namespace innfer {

// Diagnostic build only (-DINNFER_STAMPS): s_memtime stamps of one wave per workgroup, written to a
// buffer nothing else reads.  Never enabled in the shipped library; never quote its run time.
#ifdef INNFER_STAMPS
#define NSTAMP 16
__device__ unsigned long long g_stamps[8192 * NSTAMP];
#define STAMP(i)                                                                              \
    do {                                                                                      \
        if (blockIdx.x < 8192 && threadIdx.x == 0)                                            \
            g_stamps[blockIdx.x * NSTAMP + (i)] = (unsigned long long)clock64();              \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

namespace {

#ifdef INNFER_ABLATE
#define ABL_AND(x) && (x)
#else
#define ABL_AND(x)
#endif
// The B (pixel) fragments a chunk of a tap-mask kernel multiplies, in (tap column, halo row, segment) order, with the kernel rows that use each:
// the walk of the software-pipelined four-tap loop (conv3x3_pc, TM 0x1B / 0x1B0).
struct TapWalk { int n; int s[3 * 8 * 2], rr[3 * 8 * 2], seg[3 * 8 * 2]; };
constexpr TapWalk make_tap_walk(int tm, int rpw, int nseg) {
    TapWalk w{};
    for (int sc = 0; sc < 3; ++sc) {
        if (!((tm >> sc) & 0x49)) continue;
        for (int rr = 0; rr < rpw + 2; ++rr) {
            bool need = false;
            for (int r = 0; r < 3; ++r) need = need || (((tm >> (r * 3 + sc)) & 1) && rr - r >= 0 && rr - r < rpw);
            if (!need) continue;
            for (int sg = 0; sg < nseg; ++sg) { w.s[w.n] = sc; w.rr[w.n] = rr; w.seg[w.n] = sg; ++w.n; }
        }
    }
    return w;
}
template <int TM_, int RPW_, int NSEG_> struct TapWalkOf { static constexpr TapWalk value = make_tap_walk(TM_, RPW_, NSEG_); };

#include "conv3x3_kp.h"

#include "conv3x3_epilogue_slab.h"

template <int RPW, int NT, int OUTMODE>
__global__ __launch_bounds__(256, (RPW == 3 && NT == 2) ? 3 : 2) void conv3x3_mfma(const KP p) {
    constexpr int TH = 4 * RPW;
    constexpr int LH = TH + 2;
    constexpr int NPX = LH * LWP;
    constexpr int NQ = (NPX + 15) / 16;          // input DMA wave-instructions per chunk (the last one may be partly unused)
    constexpr int KQ = (NQ + 3) / 4;
    constexpr int IN_BYTES = NQ * 1024;
    constexpr int WROWS = NT * 16;
    constexpr int W_BYTES = 9 * WROWS * 64;
    constexpr int WQ = W_BYTES / 1024;           // 9*NT
    constexpr int KW = (WQ + 3) / 4;
    constexpr int MT = RPW * 2;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_in = smem;
    char* lds_w = smem + IN_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    STAMP(0);
    // ---- workgroup -> tiles ---------------------------------------------------------
    // Blocks b and b+8 share an XCD (round-robin dispatch): each XCD gets a contiguous run of the
    // tile list, so neighbouring halos and the next layer's reads of the same region meet in one L2,
    // and its workgroups walk that run round-robin.  With fewer workgroups than tiles (the default:
    // two per CU) a workgroup is PERSISTENT: it goes on to its next tile while the stores of the last
    // one drain, instead of holding its LDS and wave slots idle until they are acknowledged
    // (s_endpgm waits for them; profiles/r1/ablation_conv.txt).  Speed only, never correctness.
    const int bid = blockIdx.x, xcd = bid & 7;
    const int run_q = p.total >> 3, run_r = p.total & 7;
    const int run_start = xcd < run_r ? xcd * (run_q + 1) : run_r * (run_q + 1) + (xcd - run_r) * run_q;
    const int run_len = run_q + (xcd < run_r ? 1 : 0);
    const int slots = ((int)gridDim.x + 7 - xcd) >> 3;            // this launch's workgroups on this XCD
    const int per_img = p.tiles_x * p.tiles_y;
    const int li = lane & 15, lg = lane >> 4;
    constexpr int OOB = (int)0x80000000;

    // ---- per-lane constants: computed ONCE per (persistent) workgroup ---------------------------
    // Issue path: every DMA piece is ONE `buffer_load_dwordx4 ... offen lds` with a per-chunk SGPR
    // descriptor based at the tile's first halo pixel and a per-lane byte offset that does not depend
    // on the tile.  Lanes in the LDS row pad, and (border tiles only) lanes whose pixel lies outside
    // the image -- the conv's zero padding -- carry an offset beyond num_records: the buffer range
    // check then writes ZEROS to LDS (verified on gfx950 by the border cases of the parity tests).
    int loff[KQ];
    {
        const int ypar = p.up ? ((p.y0 - 1) & 1) : 0;           // parity of the first halo row (tile rows are even)
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const int q = wave + 4 * k;
            const int px = q * 16 + (lane >> 2);
            const int ly = px / LWP, lx = px - ly * LWP;
            const int slot = (lane & 3) ^ (((px >> 2) & 1) << 1);
            const int ry = p.up ? (ly + ypar) >> 1 : ly;        // source row / column relative to the base pixel
            const int rx = p.up ? (lx + 1) >> 1 : lx;
            loff[k] = (px < NPX && lx < LVALID) ? ((ry * p.Ws + rx) * 32 + slot * 8) * 2 : OOB;
        }
    }
    // LDS read bases.  The 16-B-slot swizzle is slot ^= 2*bit2(pixel index in LDS); with a 36-px pitch
    // bit2(row*36 + x) = bit2(x) ^ (row & 1), so there is one base per tap column s and row parity;
    // all ds_read_b128 stay base + immediate and bank-conflict free.
    const char* bbase[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int par = 0; par < 2; ++par) {                   // par: parity of rr (row inside the wave's strip)
            const int pb = wave * RPW * LWP + li + s;
            const int rowpar = ((wave * RPW) & 1) ^ par;
            bbase[s][par] = lds_in + pb * 64 + ((lg ^ (((((li + s) >> 2) & 1) ^ rowpar) << 1)) << 4);
        }
    const char* abase = lds_w + li * 64 + ((lg ^ (((li >> 2) & 1) << 1)) << 4);
    const int wvoff = lane * 16;
    // L2 prefetch plan (p.pf): LDS-DMA streams several times faster from L2 than from HBM and a
    // workgroup keeps only one chunk of LDS-DMA in flight, so while chunk c is landing / being computed
    // every thread touches one dword in (up to) two 128-B lines of the NEXT stage's halo tile -- chunk
    // c+1 of this tile, or chunk 0 of the workgroup's next tile: results discarded (2 KiB LDS scratch),
    // nothing waits for them until a whole compute phase later.
    constexpr int LPR = (LVALID * 64 + 127) / 128 + 1;        // 128-B lines per tile row (18)
    constexpr int NLINES = LH * LPR;
    int pfl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + 256 * i;
        const int r = id / LPR, l = id - r * LPR;
        pfl[i] = (p.pf && !p.up && id < NLINES) ? r * p.Ws * 64 + (l - 1) * 128 : OOB;   // relative to (row ty0-1, col tx0)
    }
    // 3-per-CU shape: no scratch (prefetch must stay off for it)
    char* lds_pf = smem + IN_BYTES + W_BYTES + ((RPW == 3 && NT == 2) ? 0 : wave * 512);
    f32x4 bias_r[NT];
    int bias_kg = -1;

    auto decode = [&](int jj, int& kg_, int& n_, int& ty0_, int& tx0_) {
        // rev: consecutive layers walk the frame in opposite directions, so a layer starts on the region
        // its predecessor touched last -- the part of the activations that is still in the 256 MiB
        // Infinity Cache (a cyclic sweep of a larger working set would never hit).
        const int lid = run_start + (p.rev ? run_len - 1 - jj : jj);
        kg_ = lid % p.KG;
        int tile = lid / p.KG;
        n_ = tile / per_img;
        tile -= n_ * per_img;
        const int ty = tile / p.tiles_x;
        ty0_ = p.y0 + ty * TH;
        tx0_ = (tile - ty * p.tiles_x) * TW;
    };
    auto is_edge = [&](int ty0_, int tx0_) {
        return ty0_ == 0 || ty0_ + TH + 1 > p.H || tx0_ == 0 || tx0_ + TW + 1 > p.W;
    };

    for (int j = bid >> 3; j < run_len; j += slots) {
    STAMP(0);
#ifdef INNFER_STAMPS
    if (blockIdx.x < 8192 && threadIdx.x == 0) g_stamps[blockIdx.x * NSTAMP + 14] = (unsigned long long)wall_clock64();
#endif
    int kg, n, ty0, tx0;
    decode(j, kg, n, ty0, tx0);
    const bool edge = is_edge(ty0, tx0);
    const char* img = (const char*)(p.in + (long)n * p.in_img_stride);
    const char* in_tile = img + ((long)((ty0 - 1) >> p.up) * p.Ws + ((tx0 >> p.up) - 1)) * 64;
    const char* pf_tile = img + ((long)(ty0 - 1) * p.Ws + tx0) * 64;         // prefetch only runs with up == 0
    int voff[KQ], pfo[2];
#pragma unroll
    for (int k = 0; k < KQ; ++k) voff[k] = loff[k];
    pfo[0] = pfl[0]; pfo[1] = pfl[1];
    if (edge) {
#pragma unroll
        for (int k = 0; k < KQ; ++k) {                   // border tiles only: recomputed, not kept in registers
            const int px = (wave + 4 * k) * 16 + (lane >> 2);
            const int ly = px / LWP, lx = px - ly * LWP;
            const int Y = ty0 - 1 + ly, X = tx0 - 1 + lx;
            if (Y < 0 || Y >= p.H || X < 0 || X >= p.W) voff[k] = OOB;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + 256 * i;
            const int r = id / LPR, l = id - r * LPR;
            const int Y = ty0 - 1 + r, lb = tx0 * 64 + (l - 1) * 128;
            if (Y < 0 || Y >= p.H || lb < 0 || lb >= p.W * 64) pfo[i] = OOB;
        }
    }
    // the workgroup's next tile, for the cross-tile prefetch (border tiles are not prefetched)
    const char* pf_next = nullptr;
    if (p.pf >= 2 && !p.up && j + slots < run_len) {
        int kg2, n2, ty2, tx2;
        decode(j + slots, kg2, n2, ty2, tx2);
        if (!is_edge(ty2, tx2))
            pf_next = (const char*)(p.in + (long)n2 * p.in_img_stride) + ((long)(ty2 - 1) * p.Ws + tx2) * 64;
    }

    // ---- accumulators start at the bias ----------------------------------------
    const int cbase = kg * WROWS + 4 * NT * lg;     // first of this lane's 4*NT channels
    if (kg != bias_kg) {
#pragma unroll
        for (int t = 0; t < NT; ++t) bias_r[t] = *(const f32x4*)(p.bias + cbase + 4 * t);
        bias_kg = kg;
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[t][m] = bias_r[t];
    const char* w_tile = (const char*)p.wpk + (long)kg * p.nchunks * W_BYTES;
    STAMP(1);

    for (int c = 0; c < p.nchunks; ++c) {
        // stage chunk c: halo tile + weight panel, straight into LDS
#if defined(__HIP_DEVICE_COMPILE__)      // the LDS buffer-load builtin only exists in the device pass
        {
            const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(in_tile + c * p.in_gbytes), 0, 0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(w_tile + (long)c * W_BYTES), 0, W_BYTES, 0x00020000);
#pragma unroll
            for (int k = 0; k < KQ; ++k) {
                const int q = wave + 4 * k;
                if (q < NQ ABL_AND(!(p.abl & 4)))
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        ri, (__attribute__((address_space(3))) void*)(lds_in + q * 1024), 16, voff[k], 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const int j = wave + 4 * k;
                if (j < WQ ABL_AND(!(p.abl & 2)))
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        rw, (__attribute__((address_space(3))) void*)(lds_w + j * 1024), 16, wvoff, j * 1024, 0, 0);
            }
        }
        asm volatile("" ::: "memory");                      // keep the prefetch loads BEHIND the DMA pieces
        const bool last = c + 1 == p.nchunks;
        const bool do_pf = p.pf && !p.up && (!last || pf_next != nullptr);
        if (do_pf) {
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(last ? pf_next : pf_tile + (c + 1) * p.in_gbytes), 0, 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (__attribute__((address_space(3))) void*)lds_pf, 4, last ? pfl[0] : pfo[0], 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (__attribute__((address_space(3))) void*)(lds_pf + 256), 4, last ? pfl[1] : pfo[1], 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // the DMA pieces, not the two prefetch loads
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#else
        (void)voff; (void)wvoff; (void)w_tile; (void)in_tile; (void)pf_tile; (void)pf_next; (void)KW; (void)pfo; (void)lds_pf;
#endif
        if (c < 3) STAMP(2 + 3 * c);
        asm volatile("s_barrier" ::: "memory");             // raw: __syncthreads() would drain vmcnt to 0
        if (c < 3) STAMP(3 + 3 * c);

#ifdef INNFER_ABLATE
        if (!(p.abl & 8))
#endif
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            f16x8 a[3][NT];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    a[r][t] = *(const f16x8*)(abase + ((r * 3 + s) * WROWS + t * 16) * 64);
#pragma unroll
            for (int rr = 0; rr < RPW + 2; ++rr) {
#pragma unroll
                for (int seg = 0; seg < 2; ++seg) {
                    const f16x8 b = *(const f16x8*)(bbase[s][rr & 1] + (rr * LWP + seg * 16) * 64);
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const int rw = rr - r;
                        if (rw >= 0 && rw < RPW) {
#pragma unroll
                            for (int t = 0; t < NT; ++t)
                                acc[t][rw * 2 + seg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                    a[r][t], b, acc[t][rw * 2 + seg], 0, 0, 0);
                        }
                    }
                }
            }
        }
        asm volatile("s_barrier" ::: "memory");
        if (c < 3) STAMP(4 + 3 * c);
    }
    STAMP(11);

    // ---- epilogue: act -> residuals -> store -----------------------------------
    if constexpr (OUTMODE == OUT_SLAB) {
#define EPI(A, B, C) epilogue_slab<RPW, NT, A, B, C, !(C)>(p, acc, n, ty0, tx0, wave, li, cbase)
        if (!p.res1) {
            if (p.act == 1) EPI(1, false, false); else if (p.act == 2) EPI(2, false, false); else EPI(0, false, false);
        } else if (!p.res2) {
            if (p.act == 1) EPI(1, true, false); else if (p.act == 2) EPI(2, true, false); else EPI(0, true, false);
        } else {
            if (p.act == 1) EPI(1, true, true); else if (p.act == 2) EPI(2, true, true); else EPI(0, true, true);
        }
#undef EPI
    } else
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int y = ty0 + wave * RPW + (m >> 1);
        const int x = tx0 + (m & 1) * 16 + li;
        if (y >= p.y1 || x >= p.W) continue;
        const long pix = ((long)n * p.H + y) * p.W + x;
        float v[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float f = acc[t][m][j];
                if (p.act == 1) f = f > 0.f ? f : 0.2f * f;
                else if (p.act == 2) f = f > 0.f ? f : 0.f;
                v[t][j] = f;
            }
        if (p.res1) {
            const f16* rp = p.res1 + (cbase >> 5) * p.res1_gstride + pix * 32 + (cbase & 31);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f16x4 r4 = *(const f16x4*)(rp + 4 * t);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[t][j] = __builtin_fmaf(v[t][j], p.s1, (float)r4[j]);
            }
        }
        if (p.res2) {
            const f16* rp = p.res2 + (cbase >> 5) * p.res2_gstride + pix * 32 + (cbase & 31);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f16x4 r4 = *(const f16x4*)(rp + 4 * t);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[t][j] = __builtin_fmaf(v[t][j], p.s2, (float)r4[j]);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) FP32_VALUE(v[t][j]);
        if constexpr (OUTMODE == OUT_NCHW) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ch = cbase + 4 * t + j;
                    if (ch < p.K) {
                        const long o = (((long)n * p.K + ch) * p.H + y) * p.W + x;
                        if (p.out_f32) ((float*)p.out)[o] = v[t][j];
                        else ((f16*)p.out)[o] = (f16)v[t][j];
                    }
                }
        } else {   // OUT_SHUFFLE2: nn.PixelShuffle(2): out[c][2y+a][2x+b] = in[4c+2a+b][y][x]
            // this lane's conv channels cbase+4t+j  <->  c = cbase/4 + t, (a,b) = (j>>1, j&1)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long opix = ((long)n * 2 * p.H + 2 * y + (j >> 1)) * (2 * p.W) + 2 * x + (j & 1);
                const int oc0 = cbase / 4;
                    f16* op = (f16*)p.out + (oc0 >> 5) * p.out_gstride + opix * 32 + (oc0 & 31);
                if constexpr (NT == 4) {
                    f16x4 h;
#pragma unroll
                    for (int t = 0; t < 4; ++t) h[t] = (f16)v[t][j];
                    *(f16x4*)op = h;
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t) op[t] = (f16)v[t][j];
                }
            }
        }
    }
#ifdef INNFER_STAMPS
    STAMP(13);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(12);
    if (blockIdx.x < 8192 && threadIdx.x == 0) g_stamps[blockIdx.x * NSTAMP + 15] = (unsigned long long)wall_clock64();   // 100 MHz
#endif
    }   // tiles of this workgroup
}


// ---------------------------------------------------------------------------------------------------
// Producer / consumer form of the same convolution (slab output).  One workgroup of 12 waves per CU:
// waves 0..7 (two per SIMD) only read LDS, issue MFMAs and run the epilogue; waves 8..11 only issue
// LDS-DMA.  LDS holds TWO stages of (halo tile + weight panel); while the consumers work on chunk g the
// loaders stage chunk g+1 (of this tile or of the workgroup's next tile), then every wave meets at one
// raw barrier per chunk.  Compared with two independent single-buffered workgroups per CU (the kernel
// above) the weight panel is staged once per CU instead of twice, the tile is 8*RPW rows tall, the
// consumers never stall on DMA issue and the staging of a chunk overlaps a full compute phase.
#ifdef INNFER_STAMPS
#define PCT(var) const unsigned long long var = clock64()
#define PCACC(slot, t1, t0) do { if (lane == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * NSTAMP + (slot)] += (t1) - (t0); } while (0)
#else
#define PCT(var) do { } while (0)
#define PCACC(slot, t1, t0) do { } while (0)
#endif
#include "conv3x3_fuse.h"

#include "conv3x3_stats_gate_rlds.h"

#include "conv3x3_planar.h"

// max(x, 0) of a pixel fragment (BRELU kernels): four v_pk_max_f16.  Inline asm: the builtin max is the IEEE maxnum -- the compiler canonicalises its operand first
// (a second v_pk_max_f16 x, x per register plus the hazard nops between the two) and the eight-instruction form cost the phase kernels 14 %.
__device__ __forceinline__ f16x8 relu_frag(f16x8 b) {
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t v = __builtin_bit_cast(u32x4_t, b);
#pragma unroll
    for (int e = 0; e < 4; ++e) { unsigned o; asm("v_pk_max_f16 %0, %1, 0" : "=v"(o) : "v"(v[e])); v[e] = o; }
    return __builtin_bit_cast(f16x8, v);
}

// S9: a 7x7 convolution as nine 3x3 convolutions over displaced copies of the input -- virtual chunk c = (sub, group): the loader reads
// channel group `group` displaced by (3*(sub/3 - 1), 3*(sub%3 - 1)) pixels and the weight panel of that chunk holds the 3x3 block
// (sub/3, sub%3) of the 7x7 kernel zero-padded to 9x9; the consumers see 9x as many chunks of an ordinary 3x3 conv.  Each input pixel
// is staged 9 times instead of 49 (gather GEMM).  Padding 3: zero, or mirrored (`reflect`).
// POLY: a dilation-d 3x3 conv (zero padding d) as ordinary 3x3 convs on the d*d polyphase components of the image ("space to batch"
// folded into the addressing): the loader's pixel steps and the epilogue's store steps are d pixels, everything else is unchanged.
// TM: tap mask (bit r*3 + s).  A 1x1 conv is the centre tap only (TM = 0x10): the weight panel then holds, the loaders stage and the consumers
// multiply ONE tap instead of nine (panels from conv_pack_taps).
// CV: image canvas.  A batch of N equally sized images (the 200 x 200 tiles of chop_forward: 7 x 32 columns and 9 x 24 rows cover 224 x 216, 21 %
// of the MFMA work on pixels that do not exist) is tiled as ONE canvas: the images are the cells of a cv_gx x cv_gy grid with a one-pixel gutter
// to the right of and below every cell (cell pitch (H + 1) x (W + 1)), and the 32-pixel-wide tile lattice runs over the whole canvas.  A gutter
// pixel is staged as zeros -- the zero padding of BOTH neighbouring images -- and never stored; a tile that straddles cells stages / stores each
// pixel at its own image's address (per-lane offsets, derived once per tile; tiles inside one cell take the plain path).  64 tiles of 200 x 200:
// 3417 canvas tiles instead of 4032.  Results are bit-identical to the per-image lattice: every output pixel sees the same operands in the same order.
// NSI: input stages in LDS.  2: halo tile + weight panel of a chunk are one stage, the loaders stage chunk g + 1 while the consumers work on chunk g
// and then WAIT for it to land before the barrier -- for that wait (a third of the loaders' time in the 32-output layers) nothing new is in
// flight.  3 (16-row tiles, 32-output layers): a ring of three input slots and two weight slots; during step g the loaders issue the weight
// panel of chunk g + 1 and the halo tile of chunk g + 2 and wait (counted vmcnt) only for what was issued a whole step earlier, so the
// request stream into HBM never pauses.
// NCW: consumer waves (8: two per SIMD; 4: one per SIMD with twice the rows each -- the same tile with 0.25 instead of 0.42 ds_read_b128 per MFMA,
// 256 registers per wave).
// TMF: bits 0..8 the tap mask of the 3x3 lattice (bit r * 3 + s; the weight panel holds the set taps in that order) + mode flags:
//   0x1FF  3x3 conv (software-pipelined fragment reads)          0x010  1x1 conv                      0x092  column taps (S9: 7 x 1 conv as three blocks)
//   0x01B  ConvTranspose2d(k, 2, 1): one output phase per channel group on a lattice shifted by the phase, scatter into the 2x slab
//   0x1B0 | 0x200  Conv2d(4, 2, 1): the loader gathers the space-to-depth source (chunk = (phase, channel group))
//   | 0x400  grids <= 16 wide: two images per tile row (PAIR)           | 0x800  (1x1) operand = LeakyReLU(running sum over the chunks) (PPON's c2)
//   | 0x1000 partial norm statistics out of the epilogue (epilogue_stats)
// Every flag is compile-time: the instantiations of the SR path (0x1FF without flags) contain none of the other modes' code.
template <int RPW, int NT, int NLW, int OUTMODE, bool S9 = false, bool POLY = false, int TMF = 0x1FF, bool CV = false, int NSI = 2, int NCW = 8>
__global__ __launch_bounds__(64 * (NCW + NLW), 1) void conv3x3_pc(const KP p) {
    // TMF: tap mask (bits 0..8) + 0x200 = the stride-2 gather loader (S2): Conv2d(k 4, s 2, p 1) as the 2x2-tap conv of the space-to-depth input --
    // chunk c is (phase (pa, pb) = c / ncg, channel group c % ncg); cell (cy, cx) of phase (pa, pb) is source pixel (2cy + pa - 1, 2cx + pb - 1)
    // (the lattice of the image padded by one pixel), so output (y, x) reads cells y, y + 1 (taps r, s in {1, 2}: mask 0x1B0) and the padding
    // is the loader's range check.  H, W: the OUTPUT grid; the source image is Hs x Ws = 2H x 2W.
    // + 0x400 (PAIR): images at most 16 pixels wide -- a tile row is TWO images side by side: LDS columns 0..17 the halo row of image 2q, 18..35 that of
    // image 2q + 1 (the 36-pixel pitch is exactly two 18-pixel halo rows), segment 0 / 1 of the MFMA walk = image 2q / 2q + 1 (the deep UNet levels;
    // until round 3 the second segment was idle).  Every output pixel sees the same operands in the same order as in a tile of its own.
    constexpr int TM = TMF & 0x1FF;
    constexpr bool S2 = (TMF & 0x200) != 0;
    constexpr bool PAIR = (TMF & 0x400) != 0;
    constexpr int NSEG = 2;
    constexpr int TWI = PAIR ? 16 : TW;                 // image columns a tile row segment set covers per image
    static_assert(!PAIR || (__builtin_popcount(TMF & 0x1FF) == 4 && !S9 && !POLY && !CV), "image pairs: the four-tap kernels");
    // + 0x800 (one-tap kernels): the B operand of chunk k is LeakyReLU(0.2) of the running sum of chunks 0 .. k of the pixel -- PPON's
    // cat(d1, d1 + d2, .., d1 + .. + d8) -> act -> c2 (PPON_arch.py:104-114) without the pass that materialises it: the consumer keeps the
    // fp32 running sums of its own pixels in registers (same additions in the same order as that pass made: same bits)
    constexpr bool PFX = (TMF & 0x800) != 0;
    constexpr bool STATS = (TMF & 0x1000) != 0;          // + 0x1000: partial norm statistics out of the epilogue (epilogue_stats)
    // + 0x2000: fp32-accurate mode on split operands.  A tensor is a pair of fp16 slabs (hi, lo = (x - hi) * 2^11; see epilogue_slab_split), a weight
    // likewise a pair of panels, and x * w = xh * wh + 2^-11 (xh * wl + xl * wh) (the 2^-22 term is dropped): the launch runs 3 * ncg virtual chunks --
    // [0, ncg) stage (xh, wl), [ncg, 2 ncg) (xl, wh), [2 ncg, 3 ncg) (xh, wh) (panels packed in that order by conv_pack_split); the accumulators start
    // at zero, are scaled by 2^-11 (exact) and receive the bias when the third part begins: ONE accumulator set, the fp16 kernel's MFMA stream.
    constexpr bool SPLIT = (TMF & 0x2000) != 0;
    static_assert(!SPLIT || (!S9 && !POLY && !S2 && !PFX && !STATS && (TM == 0x1FF || TM == 0x10)), "split operands: plain 3x3 and 1x1 convs");
    // + 0x20000 (FUSE): HR_conv0 -> conv_last in one kernel (RRDBNet_arch.py:36-42: HR_conv0 = conv + LeakyReLU, conv_last = conv, both 3x3) without recomputing a
    // halo.  The epilogue turns the tile's LeakyReLU'd result into the MFMA's B operand in registers (the k order of the last conv's panel is chosen so that a
    // lane's sixteen accumulator channels ARE its two k-step fragments), multiplies it by the 27 x 64 matrix W_last[(c, dy, dx)][k] (16 MFMAs per wave), parks
    // the 27 products per pixel in the LDS stage the tile has finished with (fp32, 64 KB), and every output pixel of the tile's 18 x 34 neighbourhood sums the
    // nine that lie inside the tile.  Pixels at least one pixel inside the tile are complete (+ bias -> planar output); the 92 pixels on the tile's rim and the
    // 100 just outside it get partial sums (fl_side), finished by conv_fuse_combine from the two to four tiles that meet there, in a fixed order.  The 4.25 GB
    // HR slab of a 1080p -> 4K frame is neither written nor read.  Two more workgroup barriers on a tile's last chunk (loaders included).
    // + 0x40000 (RLDS): the dense block's last conv (RRDBNet_arch.py:161-165: x5 = conv5(cat(x, x1 .. x4)); x5 * 0.2 + x) takes its residual x from LDS instead
    // of re-reading it from memory in the epilogue: x IS the conv's input channel groups 0 and 1, so the chunks are walked in the order 2, 3, .., 0, 1 (loader:
    // chunk c stages group (c + 2) mod nchunks and its panel) and at the end of the last two steps the live stage holds exactly the 32 residual channels half of
    // the lanes need, at the centre tap's position of their own pixels: those lanes add x / s1 to their accumulators (s1 = 0.2: x * 5, exact product, one fp32
    // rounding of the sum), the epilogue scales by s1.  265 MB of every 1327 MB launch (1080p) are no longer read twice; the RRDB-end launches, whose two
    // residuals' loads did not fit the registers as one batch, keep ONE memory residual and hoist its loads like the others.
    constexpr bool RLDS = (TMF & 0x40000) != 0;
    static_assert(!RLDS || (RPW == 2 && NT == 4 && NCW == 8 && NSI == 2 && OUTMODE == OUT_SLAB && (TMF & 0x3FFFF) == 0x1FF && !S9 && !POLY), "residual from LDS: the plain 64-channel instantiation (and its canvas form)");
    // + 0x80000 (SGATE): PAN's pixel attention behind an up-conv (PAN_arch.py:11-35: upconv -> PA: x * sigmoid(conv1x1(x)) -> LeakyReLU) inside the up-conv's
    // epilogue: a lane's accumulators are 8 consecutive channels of its pixel (NT = 2 row order), i.e. after the fp16 conversion -- the value the two-launch
    // schedule stored -- the B fragment of the 1x1 conv; two MFMAs per pixel tile against the 32 x 32 gate matrix held in registers, then v * sigmoid(g) goes
    // through the ordinary epilogue (its LeakyReLU, the store).  Same fp16 operands and the same MFMA as the two launches; the sigmoid is the fast form, so results
    // agree to the last fp16 rounding -- without the round trip of the 32-channel HR tensor (531 MB written and read again at 2160 x 3840).
    constexpr bool SGATE = (TMF & 0x80000) != 0;
    static_assert(!SGATE || (RPW == 3 && NT == 2 && NCW == 8 && OUTMODE == OUT_SLAB && ((TMF & 0x7FFFF) == 0x1FF || (TMF & 0x7FFFF) == 0x21FF) && !S9 && !POLY), "self gate: the 32-output slab kernel (its canvas form, its split-operand form)");
    // + 0x100000 (BRELU): the pixel operand is max(x, 0) of the stored slab -- the UNet keeps ONE stored form of a skip tensor (LeakyReLU, the down conv's operand);
    // the up conv that reads the concatenation applies the ReLU as it takes a fragment from LDS (4 v_pk_max_f16 per fragment, i.e. per 2 .. 8 MFMAs).
    // max(lrelu(v), 0) == relu(v) bit for bit in fp16, so the results are those of the two-view form.
    constexpr bool BRELU = (TMF & 0x100000) != 0;
    // + 0x200000 (UP4): all four output phases of a 2x transposed conv (the SR networks' up-convs as phase convs, net.hip) in ONE visit of a tile.  The
    // phase-lattice form (TM 0x1B) visits a tile once per phase and stages its input tile each time: the launch moves 4 x 128 B of input per LR pixel beside
    // 512 B of output, and the ablation (profiles/r4/upconv_bound.txt) shows the input stream is 0.9 of the 2.35 ms the frame's two up-convs take.  Here the
    // tile's two input groups (C = 64) stay in two of three LDS slots while the eight weight panels (phase, group) stream through the two weight slots:
    // step g of a workgroup is (tile g / 8, phase (g % 8) / 2, group g % 2); phase (a, b) multiplies the 2 x 2 tap block (a .. a + 1) x (b .. b + 1) of
    // the UNSHIFTED 3 x 3 halo tile (= the shifted lattice's taps {-1, 0}^2 at virtual pixel (y + a, x + b)) and stores through the phase-lattice
    // epilogue after its second group.  The loaders fetch the next tile's group 0 into the free slot at step 0 and its group 1 into the slot the
    // current tile's group 0 leaves after step 6.  Same MFMAs, same operands, same order per output value as the four-visit form: same bits.
    constexpr bool UP4 = (TMF & 0x200000) != 0;
    // + 0x400000 (ROWP): the plane row order of the 64-channel groups (see toff_slab): plain 3x3 slab convs (canvas, RLDS, FUSE forms included) and the one-pass up-conv
    constexpr bool ROWP = (TMF & 0x400000) != 0;
    static_assert(!ROWP || (RPW == 2 && NT == 4 && NCW == 8 && OUTMODE == OUT_SLAB && (TMF & 0x1FF) == 0x1FF && (TMF & ~0xE601FF) == 0 && !S9 && !POLY), "plane row order: the 64-channel slab kernels");
    // + 0x800000 (PSH): nn.PixelShuffle(2) as the store of a conv nf -> 4 nf (pixelshuffle_block, block.py:333-346; SRResNet's up stages, RRDBNet(upsample_mode=
    // 'pixelshuffle')) on THIS kernel -- until round 4 those launches ran on the two-workgroup kernel of round 1 (0.30 of the MFMA peak, a third of an SRResNet frame).
    // The panels are phase-major (conv_pack_shuffle2) in the plane row order, so a channel group is one output phase of 64 channels and the epilogue is the
    // transposed-conv phase store without the lattice shift (epilogue_slab PSH).  Same MFMAs in the same order per value as the old form: same bits.
    constexpr bool PSH = (TMF & 0x800000) != 0;
    static_assert(!PSH || (ROWP && (TMF & ~0xC001FF) == 0 && NSI == 2 && !CV), "pixel-shuffle store: the plain 64-channel plane-order instantiation");
    static_assert(!UP4 || (RPW == 2 && NT == 4 && NCW == 8 && NSI == 3 && OUTMODE == OUT_SLAB && (TMF & 0x1FFFFF) == 0x1FF && !S9 && !POLY && !CV), "one-pass phases: the 64-channel slab kernel on three input slots");
    constexpr bool FUSE = (TMF & 0x20000) != 0;
    static_assert(!FUSE || (RPW == 2 && (NT == 4 || (NT == 2 && !ROWP)) && NCW == 8 && NSI == 2 && OUTMODE == OUT_SLAB && (TMF & 0x1FFFF) == 0x1FF && !S9 && !POLY && !CV), "the fused last conv: the plain 64-channel instantiation (and the 32-channel one on 16-row tiles)");
    constexpr int TH = NCW * RPW;
    constexpr int LH = TH + 2;
    constexpr int NPX = LH * LWP;
    constexpr int NQ = (NPX + 15) / 16;
    constexpr int KQ = (NQ + NLW - 1) / NLW;
    constexpr int IN_BYTES = NQ * 1024;
    constexpr int WROWS = NT * 16;
    constexpr int NTAP = UP4 ? 4 : __builtin_popcount(TM);        // taps in the panel, in (r, s) order (UP4: a phase's 2 x 2 block)
    constexpr int W_BYTES = NTAP * WROWS * 64;
    constexpr int WQ = W_BYTES / 1024;
    constexpr int KW = (WQ + NLW - 1) / NLW;
    constexpr int MT = RPW * 2;
    constexpr int STAGE = IN_BYTES + W_BYTES;
    constexpr int OOB = (int)0x80000000;
#ifdef INNFER_NO_PIPE
    constexpr bool PIPE = false;                 // A/B build: the compiler's own placement of the fragment reads
#else
    constexpr bool PIPE = true;
#endif
#ifdef INNFER_NO_PIPE4
    constexpr bool PIPE4 = false;                // A/B build: the four-tap kernels with the compiler's placement of the fragment reads
#else
    constexpr bool PIPE4 = true;
#endif

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;

    const int bid = blockIdx.x, xcd = bid & 7;
    const int run_q = p.total >> 3, run_r = p.total & 7;
    const int run_start = xcd < run_r ? xcd * (run_q + 1) : run_r * (run_q + 1) + (xcd - run_r) * run_q;
    const int run_len = run_q + (xcd < run_r ? 1 : 0);
    const int slots = ((int)gridDim.x + 7 - xcd) >> 3;
    const int per_img = p.tiles_x * p.tiles_y;
    const int j0 = bid >> 3;
    if (j0 >= run_len) return;                                   // whole workgroup: no tiles
    const int ntiles = (run_len - j0 + slots - 1) / slots;
    const int G = ntiles * p.nchunks;                            // chunks this workgroup goes through

    // (always_inline, like the loaders' lambdas: left as a call it takes the address of the kernel-argument struct, which then lives in scratch memory)
    auto decode = [&](int jj, int& kg_, int& n_, int& ty0_, int& tx0_) __attribute__((always_inline)) {
        const int lid = run_start + (p.rev ? run_len - 1 - jj : jj);
        kg_ = lid % p.KG;
        int tile = lid / p.KG;
        n_ = tile / per_img;
        tile -= n_ * per_img;
        const int ty = tile / p.tiles_x;
        ty0_ = p.y0 + ty * TH;
        tx0_ = (tile - ty * p.tiles_x) * TW;
        if constexpr (TM == 0x1B) {
            // one phase of a 2x transposed conv per channel group: phase (a, b) of ConvTranspose2d(4, 2, 1) is the conv of taps (dy, dx) in
            // {-1, 0}^2 taken at the virtual pixel (y + a, x + b) -- the same four-tap kernel on a tile lattice shifted by (a, b)
            const int ph = kg_ * WROWS / p.phase_c;
            ty0_ += ph >> 1; tx0_ += ph & 1;
        }
    };
    // POLY: also the tile's dilation; with rate groups the channel group IS the rate and every rate has its own tile grid
    auto decode_poly = [&](int jj, int& kg_, int& n_, int& ty0_, int& tx0_, int& d_) {
        if (p.nrate == 0) { decode(jj, kg_, n_, ty0_, tx0_); d_ = p.dil; return; }
        const int lid = run_start + (p.rev ? run_len - 1 - jj : jj);
        int r = 0;
        while (r + 1 < p.nrate && lid >= p.rate_start[r + 1]) ++r;
        kg_ = r; d_ = r + 1;
        int tile = lid - p.rate_start[r];
        const int tx = ((p.fullW + d_ - 1) / d_ + TW - 1) / TW, ty = ((p.fullH + d_ - 1) / d_ + TH - 1) / TH;
        n_ = tile / (tx * ty);
        tile -= n_ * tx * ty;
        const int row = tile / tx;
        ty0_ = row * TH;
        tx0_ = (tile - row * tx) * TW;
    };

#include "conv3x3_pc_loaders.inc"

    // ================================ consumers ================================
#if defined(INNFER_CPRIO) && defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_setprio(INNFER_CPRIO);
#endif
    const int cw = wave;
    int boffs[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int pb = cw * RPW * LWP + li + s;
            const int rowpar = ((cw * RPW) & 1) ^ par;
            boffs[s][par] = pb * 64 + ((lg ^ (((((li + s) >> 2) & 1) ^ rowpar) << 1)) << 4);
        }
    // PAIR: segment 1 starts at LDS column 18 (the second image's halo row), not 16: its own swizzle term
    int boffp[PAIR ? 3 : 1][2];
    if constexpr (PAIR) {
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int pb = cw * RPW * LWP + 18 + li + s;
                const int rowpar = ((cw * RPW) & 1) ^ par;
                boffp[s][par] = pb * 64 + ((lg ^ (((((18 + li + s) >> 2) & 1) ^ rowpar) << 1)) << 4);
            }
    }
    const int aoffs = li * 64 + ((lg ^ (((li >> 2) & 1) << 1)) << 4);
    // RLDS: byte offset of the lane's 16 residual channels (two 16-byte slots; the swizzle swaps slot PAIRS, so they stay adjacent) of its pixel tile m inside a
    // stage: LDS pixel P = the centre tap's operand of output pixel (cw * RPW + m / 2, li + 16 (m % 2))
    int roffs[RLDS ? MT : 1];
    if constexpr (RLDS) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int P = (cw * RPW + (m >> 1) + 1) * LWP + li + (m & 1) * 16 + 1;
            roffs[m] = P * 64 + (((lg & 1) ^ ((P >> 2) & 1)) << 5);
            if constexpr (ROWP) roffs[m] = P * 64 + ((lg ^ (((P >> 2) & 1) << 1)) << 4);      // plane order: the ONE 16-byte slot with channels 8 lg .. 8 lg + 7 of the staged group
        }
    }
    f16x8 sgw[SGATE ? 2 : 1];
    [[maybe_unused]] f16x8 sgwl[(SGATE && SPLIT) ? 2 : 1];           // (fp32 mode: the lo fragments, 2 KB behind the hi ones)
    f32x4 sgb[SGATE ? 2 : 1];
    if constexpr (SGATE) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            sgw[t] = *(const f16x8*)(p.sg_w + (t * 64 + lane) * 8);
            if constexpr (SPLIT) sgwl[t] = *(const f16x8*)(p.sg_w + 1024 + (t * 64 + lane) * 8);
            sgb[t] = *(const f32x4*)(p.sg_bias + 8 * lg + 4 * t);
        }
    }
    // The four-tap chunk as a generic lambda over the tap mask (the one-pass up-conv, UP4, walks a different 2 x 2 block per phase): acc += panel at sw x halo tile at st
    auto four_tap = [&](auto tmc, f32x4 (&acc)[NT][2 * RPW], const char* st, const char* sw) __attribute__((always_inline)) {
        constexpr int TMX = decltype(tmc)::value;
        // The four-tap kernels (the stride-2 convs and the transposed-conv phases of the UNet / ResNet generators) with the fragment reads
        // software-pipelined like the nine-tap loop above: all sixteen weight fragments of the chunk first, then the pixel fragments through a
        // three-register ring, each read two MFMA groups ahead of its use.  Left to the scheduler, every read sat in front of its first use:
        // a chunk-step took the LDS phase PLUS the MFMA phase (1.8 us for 1.0 us of matrix work).  Same MFMAs in the same order: same bits.
        constexpr const TapWalk& WK = TapWalkOf<TMX, RPW, NSEG>::value;      // (a class-scope constant: a local copy indexed by a loop variable may land in scratch)
        constexpr int NBW = TapWalkOf<TMX, RPW, NSEG>::value.n;
        constexpr int R0 = (TMX & 0x007) ? 0 : 1, S0 = (TMX & 0x049) ? 0 : 1;          // the mask is the 2 x 2 block of taps (R0 .. R0 + 1) x (S0 .. S0 + 1)
        static_assert(TMX == (0x1B << (3 * R0 + S0)), "four taps: a 2 x 2 block of the 3 x 3 lattice");
        f16x8 a[2][NT];                                                              // the weight fragments of the current tap column, by tap row - R0
        auto lda = [&](int sc, int q) {
#pragma unroll
            for (int t = 0; t < NT; ++t) a[q][t] = *(const f16x8*)(sw + aoffs + ((q * 2 + sc - S0) * WROWS + t * 16) * 64);
        };
        f16x8 bq[3];
        auto ldb = [&](int i) {
            if constexpr (PAIR) { if (WK.seg[i]) return *(const f16x8*)(st + boffp[WK.s[i]][WK.rr[i] & 1] + WK.rr[i] * LWP * 64); }
            return *(const f16x8*)(st + boffs[WK.s[i]][WK.rr[i] & 1] + (WK.rr[i] * LWP + WK.seg[i] * 16) * 64);
        };
        lda(S0, 0); lda(S0, 1);
        bq[0] = ldb(0);
        if (NBW > 1) bq[1] = ldb(1);
        if constexpr (BRELU) bq[0] = relu_frag(bq[0]);
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            if (i + 2 < NBW) bq[(i + 2) % 3] = ldb(i + 2);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int rw = WK.rr[i] - R0 - q;
                if (rw >= 0 && rw < RPW) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[t][rw * 2 + WK.seg[i]] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q][t], bq[i % 3], acc[t][rw * 2 + WK.seg[i]], 0, 0, 0);
                }
            }
            if constexpr (BRELU) { if (i + 1 < NBW) bq[(i + 1) % 3] = relu_frag(bq[(i + 1) % 3]); }      // the NEXT group's fragment, in the shadow of this group's MFMAs
            // the next column's fragments overwrite this column's as soon as their last MFMA has been issued
            if (WK.s[i] == S0 && WK.seg[i] == NSEG - 1) {
                if (WK.rr[i] == R0 + RPW - 1) lda(S0 + 1, 0);
                if (WK.rr[i] == R0 + RPW) lda(S0 + 1, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#include "conv3x3_pc_up4.inc"
    int islot = 0;                                                // NSI == 3: g % 3
    f32x4 bias_r[NT];
    int bias_kg = -1;
    f32x4 acc[NT][MT];
    float pfx[PFX ? MT : 1][8];          // (cleared after a tile's last chunk, in front of its epilogue: live sums there would cost the epilogue's registers)
#pragma unroll
    for (int m = 0; m < (PFX ? MT : 1); ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) pfx[m][e] = 0.f;
    int jt = j0, c = 0;
    int kg = 0, n = 0, ty0 = 0, tx0 = 0, cbase = 0, dcur = 1;
    asm volatile("s_barrier" ::: "memory");                       // chunk 0 of the first tile has landed
#ifdef INNFER_STAMPS
    const unsigned long long k_c0 = clock64(), k_w0 = wall_clock64();
#endif
    for (int g = 0; g < G; ++g) {
        if (c == 0) {
            if constexpr (POLY) decode_poly(jt, kg, n, ty0, tx0, dcur);
            else decode(jt, kg, n, ty0, tx0);
            cbase = kg * WROWS + lane_cbase<NT, ROWP>(lg);
            if (kg != bias_kg) {
#pragma unroll
                for (int t = 0; t < NT; ++t) bias_r[t] = *(const f32x4*)(p.bias + cbase + toff_lin<NT, ROWP>(t));
                bias_kg = kg;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[t][m] = SPLIT ? f32x4{0.f, 0.f, 0.f, 0.f} : bias_r[t];
        }
        if constexpr (SPLIT) {
            if (c == 2 * p.ncg) {                    // the two cross terms are complete: scale them down (exact) and go on with xh * wh on top of the bias
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[t][m][j] = __builtin_fmaf(acc[t][m][j], SPLIT_DOWN, bias_r[t][j]);
            }
        }
        const char* st = NSI == 3 ? smem + islot * IN_BYTES : smem + (g & 1) * STAGE;                       // halo tile
        const char* sw = NSI == 3 ? smem + 3 * IN_BYTES + (g & 1) * W_BYTES : st + IN_BYTES;              // weight panel
        if constexpr (NSI == 3) islot = islot == 2 ? 0 : islot + 1;
        PCT(c0);
#ifdef INNFER_ABLATE
        const bool abl_no_mfma = (p.abl & 8) != 0;
#else
        constexpr bool abl_no_mfma = false;
#endif
        if (abl_no_mfma) {
        } else if constexpr (TM == 0x1FF && PIPE && !UP4) {
            // Software-pipelined fragment reads (the nine-tap kernels).  The B fragments of a chunk are walked in (s, rr, seg) order through
            // a three-register ring, each read issued two MFMA groups (>= 8 MFMAs = 128 pipe cycles) ahead of its use; the weight fragments
            // of tap column s + 1 overwrite those of column s as soon as their last MFMA has been issued (A(s,0,*) after row RPW - 1,
            // A(s,1,*) after row RPW, A(s,2,*) at the end of the column, needed again from row 2 on).  sched_barrier pins the order: left to
            // itself the scheduler sinks every ds_read to just in front of its first use and the MFMA pipe then waits out the LDS latency
            // at each of them, both waves of the SIMD at the same points (they run in step between the chunk barriers).  Same MFMAs in the
            // same order: results are unchanged.
            constexpr int GPS = 2 * (RPW + 2);                     // B fragments (= MFMA groups) per tap column
            constexpr int NB = 3 * GPS;
            f16x8 a[3][NT];
            f16x8 bq[3];
            auto lda = [&](int sc, int r) {
#pragma unroll
                for (int t = 0; t < NT; ++t) a[r][t] = *(const f16x8*)(sw + aoffs + ((r * 3 + sc) * WROWS + t * 16) * 64);
            };
            auto ldb = [&](int i) {
                const int sc = i / GPS, j = i - sc * GPS, rr = j >> 1, seg = j & 1;
                return *(const f16x8*)(st + boffs[sc][rr & 1] + (rr * LWP + seg * 16) * 64);
            };
            lda(0, 0); lda(0, 1); lda(0, 2);
            bq[0] = ldb(0); bq[1] = ldb(1);
            if constexpr (BRELU) bq[0] = relu_frag(bq[0]);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int sc = i / GPS, j = i - sc * GPS, rr = j >> 1, seg = j & 1;
                if (i + 2 < NB) bq[(i + 2) % 3] = ldb(i + 2);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int rw = rr - r;
                    if (rw >= 0 && rw < RPW) {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[t][rw * 2 + seg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[r][t], bq[i % 3], acc[t][rw * 2 + seg], 0, 0, 0);
                    }
                }
                if constexpr (BRELU) { if (i + 1 < NB) bq[(i + 1) % 3] = relu_frag(bq[(i + 1) % 3]); }      // the NEXT group's fragment, in the shadow of this group's MFMAs
                if (sc < 2 && seg == 1) {
                    if (rr == RPW - 1) lda(sc + 1, 0);
                    if (rr == RPW) lda(sc + 1, 1);
                    if (rr == RPW + 1) lda(sc + 1, 2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (NTAP == 4 && !PFX && PIPE4 && RPW <= 6 && !UP4) {
            four_tap(std::integral_constant<int, TM>{}, acc, st, sw);
        } else {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (!((TM >> s) & 0x49)) continue;                     // no tap in this column (loop constants: folded at compile time)
            f16x8 a[3][NT];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if ((TM >> (r * 3 + s)) & 1)                   // panel slot = rank of the tap among the mask's set bits
                        a[r][t] = *(const f16x8*)(sw + aoffs + (__builtin_popcount(TM & ((1 << (r * 3 + s)) - 1)) * WROWS + t * 16) * 64);
#pragma unroll
            for (int rr = 0; rr < RPW + 2; ++rr) {
                bool need = false;
#pragma unroll
                for (int r = 0; r < 3; ++r) need = need || (((TM >> (r * 3 + s)) & 1) && rr - r >= 0 && rr - r < RPW);
                if (!need) continue;
#pragma unroll
                for (int seg = 0; seg < NSEG; ++seg) {
                    f16x8 b = *(const f16x8*)(st + ((PAIR && seg) ? boffp[PAIR ? s : 0][rr & 1] + rr * LWP * 64 : boffs[s][rr & 1] + (rr * LWP + seg * 16) * 64));
                    if constexpr (BRELU) b = relu_frag(b);
                    if constexpr (PFX) {           // (one tap: row rr feeds output row rr - 1 only)
                        static_assert(!PFX || TM == 0x10, "the running-sum operand belongs to the one-tap kernels");
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float& run = pfx[PFX ? (rr - 1) * 2 + seg : 0][e];
                            run += (float)b[e];
                            b[e] = (f16)fmaxf(run, 0.2f * run);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const int rw = rr - r;
                        if (((TM >> (r * 3 + s)) & 1) && rw >= 0 && rw < RPW) {
#pragma unroll
                            for (int t = 0; t < NT; ++t)
                                acc[t][rw * 2 + seg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                    a[r][t], b, acc[t][rw * 2 + seg], 0, 0, 0);
                        }
                    }
                }
            }
        }
        }
        if constexpr (RLDS) {
            // chunk nchunks - 2 is input group 0 = residual channels 0..31 (lanes lg 0, 1), chunk nchunks - 1 group 1 = channels 32..63 (lanes lg 2, 3)
            if constexpr (ROWP) {        // plane order: every lane takes 8 channels from each of the two groups -- tiles 0, 1 from group 0, tiles 2, 3 from group 1
                if (c == p.nchunks - 2) residual_from_lds_plane<MT, 0>(acc, st, roffs, p.rs1);
                else if (c == p.nchunks - 1) residual_from_lds_plane<MT, 1>(acc, st, roffs, p.rs1);
            } else {
                if (c >= p.nchunks - 2 && (lg >> 1) == c - (p.nchunks - 2)) residual_from_lds<MT>(acc, st, roffs, p.rs1);
            }
        }
        PCT(c1);
        if (cw == 0) PCACC(0, c1, c0);
        if (++c == p.nchunks) {
            c = 0;
            jt += slots;
            if constexpr (SGATE && SPLIT) self_gate_split<MT>(acc, sgw, sgwl, sgb);
            else if constexpr (SGATE) self_gate<MT>(acc, sgw, sgb);
            if constexpr (PFX) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int e = 0; e < 8; ++e) pfx[m][e] = 0.f;
            }
#include "conv3x3_pc_tile_end.inc"
        }
        PCT(c2);
#ifdef INNFER_ABLATE
        if ((p.abl & 32) && c != 0) continue;                              // free-running loaders: the consumers meet them once per tile (c was advanced above)
#endif
        asm volatile("s_barrier" ::: "memory");
        PCT(c3);
        if (cw == 0) { PCACC(1, c2, c1); PCACC(2, c3, c2); PCACC(6, 1, 0); }
    }
#ifdef INNFER_STAMPS
    if (cw == 0) { PCACC(7, clock64(), k_c0); PCACC(8, wall_clock64(), k_w0); }
#endif
}

template <int RPW, int NT, int NLW, int OUTMODE = OUT_SLAB, bool S9 = false, bool POLY = false, int TM = 0x1FF, bool CV = false, int NSI = 2, int NCW = 8>
int launch_pc(const KP& kp, int N, hipStream_t s);

// Per-device state (a process may drive several GPUs): CU count, and which devices already carry a kernel's
// dynamic-LDS attribute (function attributes belong to the device's copy of the code object).
int current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess && dev >= 0 ? dev : 0;
}

int num_cus() {
    static int n[64] = {};
    const int dev = current_device() & 63;
    if (!n[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n[dev] = v;
    }
    return n[dev];
}

template <typename F>
int ensure_lds_attr(F* kernel, int lds_bytes, unsigned long long& done_mask) {
    const unsigned long long bit = 1ull << (current_device() & 63);
    if (!(done_mask & bit)) {
        INNFER_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        done_mask |= bit;
    }
    return INNFER_OK;
}


template <int RPW, int NT, int OUTMODE>
int launch_t(const KP& kp, int N, hipStream_t s) {
    constexpr int TH = 4 * RPW;
    constexpr int LDS = (((TH + 2) * LWP + 15) / 16) * 1024 + 9 * NT * 16 * 64 + ((RPW == 3 && NT == 2) ? 0 : 2048);   // + prefetch scratch
    static unsigned long long attr_done = 0;
    if (int rc = ensure_lds_attr(conv3x3_mfma<RPW, NT, OUTMODE>, LDS, attr_done)) return rc;
    KP k = kp;
    k.tiles_x = (k.W + TW - 1) / TW;
    k.tiles_y = (k.y1 - k.y0 + TH - 1) / TH;
    {
        const int pf = INNFER_KNOB("INNFER_PREFETCH", 0);   // 1: next chunk, 2: + next tile
        k.pf = (RPW == 3 && NT == 2) ? 0 : pf;
#ifdef INNFER_ABLATE
        k.abl = getenv("INNFER_ABL") ? atoi(getenv("INNFER_ABL")) : 0;     // read per launch: scripts/ablate.py changes it between runs
#endif
    }
    const long total = (long)N * k.tiles_x * k.tiles_y * k.KG;
    if (total <= 0) return INNFER_OK;
    if (total > 0x7fffffffL) return set_error(INNFER_ERR_INVALID, "conv grid too large");
    k.total = (int)total;
    const int persist = INNFER_KNOB("INNFER_PERSIST", 2);   // workgroups per CU, 0 = one per tile
    const long slots = (long)((persist == 2 && RPW == 3 && NT == 2) ? 3 : persist) * num_cus();   // = workgroups resident per CU
    const long grid = (persist > 0 && total > slots) ? slots : total;
    hipLaunchKernelGGL((conv3x3_mfma<RPW, NT, OUTMODE>), dim3((unsigned)grid), dim3(256), LDS, s, k);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

// Image canvas for a batch (conv3x3_pc<.., CV>): the grid (columns x rows of cells) with the fewest tiles, or 0 columns when the per-image lattice
// is at least as good (one image, images that already are whole tiles) or the form does not apply.
template <int TH>
int canvas_grid(const KP& k, int N, int* gy, long* tiles) {
    const long plain = (long)N * ((k.W + TW - 1) / TW) * ((k.H + TH - 1) / TH);
    *tiles = plain; *gy = 0;
    if (N < 2 || k.up || k.reflect || (k.act >= 3 && k.act != 7) || k.y0 != 0 || k.y1 != k.H || k.H < TH + 2 || k.W < LVALID || k.nrate || k.dil > 1 || k.stats_part ||
        (long)N * k.H * k.W * 64 >= 0x7fffffffL)
        return 0;
    int best = 0;
    for (int gx = 1; gx <= N; ++gx) {
        const int g_y = (N + gx - 1) / gx;
        const long t = (long)((gx * (k.W + 1) + TW - 1) / TW) * ((g_y * (k.H + 1) + TH - 1) / TH);
        if (t < *tiles) { *tiles = t; best = gx; *gy = g_y; }
    }
    return *tiles * 100 <= plain * 98 ? best : 0;         // worth it from 2 % fewer tiles
}

template <int RPW, int NT, int NLW, int OUTMODE, bool S9, bool POLY, int TM, bool CV, int NSI, int NCW>
int launch_pc(const KP& kp, int N, hipStream_t s) {
    constexpr int TH = NCW * RPW;
    constexpr int LDS = NSI * ((((TH + 2) * LWP + 15) / 16) * 1024) + 2 * (((TM & 0x200000) ? 4 : __builtin_popcount(TM & 0x1FF)) * NT * 16 * 64) + ((TM & 0x20000) ? 4096 : 0) + ((TM & 0x200000) ? 1024 : 0);      // (UP4: + the four phases' biases)
    static_assert(LDS <= 160 * 1024, "the stages must fit the CU's LDS");
    static_assert(NSI == 2 || (NSI == 3 && !S9 && !POLY), "the three-slot input ring exists for the plain and the canvas loader");
    if constexpr (OUTMODE == OUT_SLAB && !S9 && !POLY && (TM & ~0x4C2000) == 0x1FF && !CV) {      // a batch of images whose size is not a whole number of tiles
        int gy = 0; long t = 0;
        const int gx = INNFER_KNOB("INNFER_CANVAS", 1) ? canvas_grid<TH>(kp, N, &gy, &t) : 0;
        if (gx > 0) {
            KP kc = kp;
            kc.cv_gx = gx; kc.cv_gy = gy; kc.cv_h1 = kp.H + 1; kc.cv_w1 = kp.W + 1;
            return launch_pc<RPW, NT, NLW, OUTMODE, false, false, TM, true, NSI, NCW>(kc, N, s);
        }
    }
    static unsigned long long attr_done = 0;
    if (int rc = ensure_lds_attr(conv3x3_pc<RPW, NT, NLW, OUTMODE, S9, POLY, TM, CV, NSI, NCW>, LDS, attr_done)) return rc;
    KP k = kp;
    k.tiles_x = (k.W + TW - 1) / TW;
    k.tiles_y = (k.y1 - k.y0 + TH - 1) / TH;
    k.N = N;
#ifdef INNFER_ABLATE
    k.abl = getenv("INNFER_ABL") ? atoi(getenv("INNFER_ABL")) : 0;     // read per launch: scripts/ablate.py changes it between runs
#endif
    long total = (long)((TM & 0x400) ? (N + 1) / 2 : N) * k.tiles_x * k.tiles_y * k.KG;      // (0x400: two images per tile row)
    if constexpr (CV) {                       // one canvas instead of N images
        k.tiles_x = (k.cv_gx * k.cv_w1 + TW - 1) / TW;
        k.tiles_y = (k.cv_gy * k.cv_h1 + TH - 1) / TH;
        total = (long)k.tiles_x * k.tiles_y * k.KG;
    }
    if (POLY && k.nrate > 0) {               // N = images here; rate g has N * g^2 sub-images with their own tile grid
        total = 0;
        for (int g = 0; g < k.nrate; ++g) {
            const int d = g + 1;
            k.rate_start[g] = (int)total;
            total += (long)N * d * d * (((k.fullW + d - 1) / d + TW - 1) / TW) * (((k.fullH + d - 1) / d + TH - 1) / TH);
        }
        if (total <= 0x7fffffffL) k.rate_start[k.nrate] = (int)total;
    }
    if (total <= 0) return INNFER_OK;
    if (total > 0x7fffffffL) return set_error(INNFER_ERR_INVALID, "conv grid too large");
    k.total = (int)total;
    const long grid = total < num_cus() ? total : num_cus();
    hipLaunchKernelGGL((conv3x3_pc<RPW, NT, NLW, OUTMODE, S9, POLY, TM, CV, NSI, NCW>), dim3((unsigned)grid), dim3(64 * (NCW + NLW)), LDS, s, k);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

}  // namespace

}  // namespace innfer
#ifdef INNFER_STAMPS
namespace innfer { __global__ void k_clear_stamps() { for (int i = threadIdx.x; i < 8192 * NSTAMP; i += blockDim.x) g_stamps[i] = 0; } }
#endif
extern "C" int innfer_debug_clear_stamps() {
#ifdef INNFER_STAMPS
    hipLaunchKernelGGL(innfer::k_clear_stamps, dim3(1), dim3(1024), 0, 0);
    return hipDeviceSynchronize() == hipSuccess ? 0 : -2;
#else
    return 0;
#endif
}
extern "C" int innfer_debug_read_stamps(unsigned long long* h, int n_words) {
#ifdef INNFER_STAMPS
    return hipMemcpyFromSymbol(h, HIP_SYMBOL(innfer::g_stamps), (size_t)n_words * 8) == hipSuccess ? NSTAMP : -2;
#else
    (void)h; (void)n_words;
    return 0;
#endif
}
namespace innfer {

#include "conv3x3_pack.h"

static const char* conv_family(const ConvLaunch& L, double* flops, double* bytes) {
    const int y1 = L.y1 > 0 ? L.y1 : L.H;
    const double px = (double)L.N * (y1 - L.y0) * L.W;                       // pixels of the kernel's grid (ConvTranspose: the input grid, 4 phases each)
    double taps = 9, in_px = px, obytes = L.out_mode == OUT_NCHW ? (L.out_u8 ? 1.0 : L.out_f32 ? 4.0 : 2.0) : 2.0;
    const char* name = L.out_mode == OUT_NCHW ? "conv3x3_pc 3x3, planar output" : (conv_nt_for(L.K) == 4 ? "conv3x3_pc 3x3, 64-channel tiles" : "conv3x3_pc 3x3, 32-channel tiles");
    if (L.conv1x1) { taps = 1; name = "conv3x3_pc 1x1"; }
    if (L.up) in_px = px / 4;
    if (L.stride2) { taps = 16; in_px = 4 * px; name = "conv3x3_pc Conv2d(4,2,1), stride-2 loader"; }
    if (L.deconv_phases) { taps = 4; name = "conv3x3_pc ConvTranspose2d(k,2,1), phase lattice"; }
    if (L.conv7v) { taps = 7; name = "conv3x3_pc 7x1 column conv"; }
    if (L.conv7) { taps = 49; name = "conv3x3_pc 7x7 (nine displaced 3x3)"; }
    if (L.dilation > 1 || L.dilation_groups) name = "conv3x3_pc dilated 3x3 (polyphase)";
    if (L.out_mode == OUT_NCHW && L.phase_c > 0) { taps = 4; name = "conv3x3_pc outermost ConvTranspose2d(4,2,1): 4 phases, tanh, planar"; }      // (structural zeros not counted)
    if (L.split) name = "conv3x3_pc 3x3, fp32 mode (hi/lo pairs)";
    *flops = 2.0 * taps * L.K * L.C * px;
    *bytes = in_px * L.C * 2.0 + px * L.K * obytes + (L.res1 ? px * L.K * 2.0 : 0.0) + (L.res2 ? px * L.K * 2.0 : 0.0) + taps * L.K * L.C * 2.0;
    if (L.split) *bytes = 2.0 * *bytes + taps * L.K * L.C * 2.0;
    return name;
}

#ifndef UNET_NSI
#define UNET_NSI 3
#endif
int conv_launch(const ConvLaunch& L, hipStream_t s) {
    double gt_flops = 0, gt_bytes = 0;
    const char* gt_name = gt_on() ? conv_family(L, &gt_flops, &gt_bytes) : "";
    GtScope gt(s, gt_name, gt_flops, gt_bytes);
    if (L.C <= 0 || L.C % 32) return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: C=%d must be a multiple of 32", L.C);
    if (L.up && ((L.H | L.W) & 1)) return set_error(INNFER_ERR_INVALID, "conv3x3: upsampled size must be even");
    KP k{};
    k.up = L.up ? 1 : 0;
    k.H = L.H; k.W = L.W; k.Hs = L.H >> k.up; k.Ws = L.W >> k.up;
    k.in = L.in; k.in_gbytes = L.in_gstride * 2; k.in_img_stride = (long)k.Hs * k.Ws * 32;
    k.nchunks = L.C / 32;
    k.ncg = L.C / 32;
    k.wpk = L.wpk; k.bias = L.bias;
    k.out = L.out; k.out_gstride = L.out_gstride; k.out_coff = L.out_coff;
    k.K = L.K; k.KG = conv_groups(L.K);
    if ((L.act == 4 || L.act == 5) && (L.out_mode != OUT_SLAB || !L.res1 || L.res2))
        return set_error(INNFER_ERR_INVALID, "conv3x3: the gate epilogue multiplies res1 (slab output, no second residual)");
    if (L.act == 6 && L.out_mode != OUT_NCHW) return set_error(INNFER_ERR_INVALID, "conv3x3: sigmoid is a planar-output activation");
    if (L.act == 7 && (L.out_mode != OUT_SLAB || conv_nt_for(L.K) != 4 || L.K % 64 || L.res1 || L.res2 || L.conv1x1 || L.dilation > 1 || L.dilation_groups ||
                       L.deconv_phases || L.stride2 || L.conv7v || L.stats_part))
        return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: the pair gate is an epilogue of the plain 64-row slab conv (32 gated outputs per 64 rows)");
    if (L.act < 0 || L.act > 7 || (L.phase_c > 0 && (L.K % L.phase_c || L.K / L.phase_c != 4)))
        return set_error(INNFER_ERR_INVALID, "conv3x3: act=%d phase_c=%d K=%d", L.act, L.phase_c, L.K);
    k.act = L.act;
    k.res1 = L.res1; k.res1_gstride = L.res1_gstride; k.s1 = L.s1;
    k.res2 = L.res2; k.res2_gstride = L.res2_gstride; k.s2 = L.s2;
    k.y0 = L.y0; k.y1 = L.y1 > 0 ? L.y1 : L.H;
    if (k.y0 < 0 || k.y1 > L.H || k.y0 >= k.y1) return set_error(INNFER_ERR_INVALID, "conv3x3: bad row range [%d,%d)", k.y0, k.y1);
    k.out_f32 = L.out_f32;
    if (L.out_u8 && (L.out_mode != OUT_NCHW || L.K > 4 || L.phase_c > 0 || L.res1 || L.res2 || L.conv7))
        return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: the uint8 image epilogue belongs to planar outputs of <= 4 channels");
    k.out_u8 = L.out_u8; k.out_denorm = L.out_denorm; k.out_round16 = L.out_round16;
    k.outm = L.outm;
    k.rev = L.rev ? 1 : 0;
    k.phase_c = (L.out_mode == OUT_NCHW || L.deconv_phases) ? L.phase_c : 0;
    k.stats_part = L.stats_part; k.stats_cn = L.deconv_phases ? L.phase_c : L.K;
    if (L.stats_part && (L.out_mode != OUT_SLAB || conv_nt_for(L.K) != 4 || L.act || L.res1 || L.res2 || L.conv1x1 || L.dilation > 1 || L.dilation_groups))
        return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: epilogue statistics are built for the 64-channel slab kernels without activation / residual");
    k.reflect = L.reflect == 2 ? 2 : (L.reflect ? 1 : 0);
    if (L.reflect == 2 && (L.conv7 || L.out_mode != OUT_SLAB)) return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: replication padding is built for 3x3 slab convs");
    if (L.reflect && (L.up || L.H < ((L.conv7 || L.conv7v) ? 4 : (L.reflect == 2 ? 1 : 2)) || L.W < (L.conv7 ? 4 : (L.reflect == 2 ? 1 : 2))))
        return set_error(INNFER_ERR_INVALID, "conv3x3: reflection padding needs >= 2x2 pixels and no upsampled input");
    const int nt = conv_nt_for(L.K);
    if (L.out_mode != OUT_NCHW && L.K % (16 * nt))
        return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: K=%d must be a multiple of %d for slab output", L.K, 16 * nt);
    const int rpw64 = INNFER_KNOB("INNFER_RPW64", 3);
    const int rpw32 = INNFER_KNOB("INNFER_RPW32", 5);
    const int pc = INNFER_KNOB("INNFER_PC", 1);     // producer / consumer kernel for slab outputs
    if (L.rowp && !(L.rowp == 2 && L.out_mode == OUT_SHUFFLE2) && (nt != 4 || !pc || L.out_mode != OUT_SLAB || L.split || L.stats_part || L.stride2 || L.conv1x1 || L.conv7 || L.conv7v || L.prefix_lrelu ||
                   L.gate_w || L.act > 2 || L.dilation > 1 || L.dilation_groups || (L.out_coff & 31)))
        return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: the plane row order (rowp) belongs to plain 3x3 slab convs and transposed-conv phases with 64-channel output groups");
    if (L.outm && (L.out_mode != OUT_NCHW || !pc || nt != 1 || L.res1 || L.res2 || L.outm < 0 || L.outm > 4))
        return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: outm belongs to the planar last conv (<= 16 channels)");
    if (L.split) {           // fp32-accurate mode on (hi, lo) slab pairs: 3 * C / 32 virtual chunks (conv3x3_pc<.., TMF | 0x2000>)
        if (!pc || (L.act > 2 && !((L.act == 3 || L.act == 6) && L.out_mode == OUT_NCHW) && !((L.act == 4 || L.act == 5) && L.conv1x1 && nt == 2)) || L.reflect || L.dilation > 1 || L.dilation_groups || L.deconv_phases || L.stride2 || L.conv7v || L.conv7 || L.stats_part ||
            L.prefix_lrelu || L.phase_c || (long)3 * L.C / 32 > 0x7fff)
            return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3 (fp32 mode): plain 3x3 / 1x1 convs with act 0..2 (the 32-channel 1x1 conv also the PA gate), residuals, upsampled input");
        k.nchunks = 3 * k.ncg;
        k.in_lo_bytes = L.in_lo * 2; k.out_lo = L.out_lo; k.res1_lo = L.res1_lo; k.res2_lo = L.res2_lo;
        if (L.out_mode == OUT_SLAB && L.conv1x1 && (nt == 2 || nt == 4))
            return nt == 4 ? launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x2010, false, 3>(k, L.N, s) : launch_pc<2, 2, 4, OUT_SLAB, false, false, 0x2010, false, 3>(k, L.N, s);
        if (L.conv1x1) return set_error(INNFER_ERR_UNSUPPORTED, "conv1x1 (fp32 mode): slab outputs of 32- / 64-channel tiles");
        if (L.gate_w) {      // out = act(v * sigmoid(W v + b)) on the conv's fp32 result: gate_w = the hi | lo fragments of conv_pack_selfgate (4 KB)
            if (L.out_mode != OUT_SLAB || nt != 2 || L.K != 32 || L.res1 || L.res2 || L.act > 2)
                return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3 (fp32 mode): the self gate belongs to 32-output slab convs without residuals");
            k.sg_w = L.gate_w; k.sg_bias = L.gate_bias;
            return launch_pc<3, 2, 4, OUT_SLAB, false, false, 0x821FF>(k, L.N, s);
        }
        if (L.out_mode == OUT_SLAB && nt == 2) return launch_pc<3, 2, 4, OUT_SLAB, false, false, 0x21FF>(k, L.N, s);
        if (L.out_mode == OUT_SLAB && nt == 4) return launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x21FF>(k, L.N, s);
        if (L.out_mode == OUT_NCHW && nt == 1 && !L.res1 && !L.res2) return launch_pc<3, 1, 4, OUT_NCHW, false, false, 0x21FF>(k, L.N, s);
        return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3 (fp32 mode): slab outputs of 32- / 64-channel tiles or a planar output of <= 16 channels (K=%d)", L.K);
    }
    if (L.dilation_groups > 0) {   // K = 32 * groups: output channel group g (its own 32-output panel) is the conv of dilation g + 1
        if (!pc || L.out_mode != OUT_SLAB || L.K != 32 * L.dilation_groups || L.dilation_groups > 8 || L.res1 || L.res2 || L.up || L.reflect ||
            L.y0 != 0 || k.y1 != L.H || (long)L.H * L.W * 64 >= 0x7fffffffL)
            return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: dilation groups are 32-output slab convs (K = 32 * groups <= 256), no residual / upsampling");
        k.nrate = L.dilation_groups; k.KG = L.dilation_groups; k.dil = 1; k.fullH = L.H; k.fullW = L.W;
        k.y0 = 0; k.y1 = L.H;
        return launch_pc<3, 2, 4, OUT_SLAB, false, true>(k, L.N, s);
    }
    if (L.dilation > 1) {  // dilated 3x3 conv (zero padding = dilation) on the polyphase components: 32-output slab tiles, plain epilogue
        const int d = L.dilation;
        if (!pc || L.out_mode != OUT_SLAB || nt != 2 || L.res1 || L.res2 || L.up || L.reflect || L.y0 != 0 || k.y1 != L.H ||
            (long)L.H * L.W * 64 >= 0x7fffffffL || d > 64)
            return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: dilated convs are 32-output slab convs without residual / upsampling, images below 33 M pixels");
        k.dil = d; k.fullH = L.H; k.fullW = L.W;
        k.H = (L.H + d - 1) / d; k.W = (L.W + d - 1) / d; k.Hs = k.H; k.Ws = k.W;
        k.y0 = 0; k.y1 = k.H;
        return launch_pc<3, 2, 4, OUT_SLAB, false, true>(k, L.N * d * d, s);
    }
    if (L.deconv_phases) {   // ConvTranspose2d(4, 2, 1): K = 4 * phase_c, group g = phase g / (phase_c / 64); H, W: the input grid; output slab 2H x 2W
        if (!pc || L.out_mode != OUT_SLAB || nt != 4 || L.phase_c <= 0 || L.phase_c % 64 || L.K != 4 * L.phase_c || L.res1 || L.res2 || L.up || L.reflect ||
            L.act > 2 || L.y0 != 0 || k.y1 != L.H || L.dilation > 1 || L.dilation_groups)
            return set_error(INNFER_ERR_UNSUPPORTED, "deconv phases: slab output, 64-channel phase groups, no residual / upsampling / padding modes");
        // (in_relu on the phase lattice was built and measured -- scripts/r4/unet_one_view.sh: the four v_pk_max_f16 per fragment share the issue port with the MFMAs of
        //  these matrix-bound launches, +7.7 % on a 130-us launch against a 6 .. 12 us shorter post pass; only the HBM-bound outermost layer keeps the operand ReLU)
        if (L.in_relu) return set_error(INNFER_ERR_UNSUPPORTED, "deconv phases: in_relu is built for the planar <= 16-output kernel only");
        if (L.rowp && (L.C != 64 || L.phase_c != 64 || L.stats_part || L.W <= 16 || L.deconv_phases == 2))
            return set_error(INNFER_ERR_UNSUPPORTED, "deconv phases: plane-order panels belong to the one-visit form (C = 64, 64-channel phases, grids wider than 16, no statistics)");
        if (L.rowp) {
            // the SR networks' up-convs (64 -> 64): all four phases in one visit of a tile, the input tile staged once (conv3x3_pc<.., TMF | 0x200000>), panels in the
            // plane row order; lane-contiguous panels (rowp 0) run the one-phase-per-visit form below
            k.KG = 1; k.nchunks = 8;
            return launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x6001FF, false, 3>(k, L.N, s);
        }
        if (L.stats_part) return L.W <= 16 ? launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x141B, false, UNET_NSI>(k, L.N, s) : launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x101B, false, UNET_NSI>(k, L.N, s);
        return L.W <= 16 ? launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x41B, false, UNET_NSI>(k, L.N, s) : launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x1B, false, UNET_NSI>(k, L.N, s);
    }
    if (L.in_relu) {         // operand = max(stored, 0) (conv3x3_pc<.., TMF | 0x100000>): the planar 16-output kernel -- the UNet's outermost transposed conv
        if (!(pc && L.out_mode == OUT_NCHW && nt == 1 && !L.res1 && !L.res2 && !L.conv7 && !L.stride2 && !L.conv7v && !L.conv1x1 && !L.gate_w && !L.fuse_w && !L.up && !L.reflect))
            return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: in_relu is built for the planar <= 16-output kernel");
        return launch_pc<3, 1, 4, OUT_NCHW, false, false, 0x1001FF>(k, L.N, s);
    }
    if (L.stride2) {         // Conv2d(k 4, s 2, p 1): H x W = the OUTPUT grid, source image 2H x 2W; panels from conv_pack_taps(K, 4C, 0x1B0), virtual channel
                             // (2 pa + pb) * C + ci, tap (1 + dy, 1 + dx) = w[co][ci][2 dy + pa][2 dx + pb]
        if (!pc || L.out_mode != OUT_SLAB || (nt != 4 && nt != 2) || L.res1 || L.res2 || L.up || L.reflect || L.act > 2 || L.y0 != 0 || k.y1 != L.H ||
            L.dilation > 1 || L.dilation_groups || (long)L.H * L.W * 4 * 64 >= 0x7fffffffL || (nt == 2 && L.stats_part))
            return set_error(INNFER_ERR_UNSUPPORTED, "stride-2 conv: slab output of 32- / 64-channel tiles, no residual / upsampling / padding modes, sources below 33 M pixels");
        k.Hs = 2 * L.H; k.Ws = 2 * L.W; k.in_img_stride = (long)k.Hs * k.Ws * 32;
        k.nchunks = 4 * k.ncg;
        if (nt == 2) return launch_pc<3, 2, 4, OUT_SLAB, false, false, 0x3B0>(k, L.N, s);
        if (L.stats_part) return L.W <= 16 ? launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x17B0, false, UNET_NSI>(k, L.N, s) : launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x13B0, false, UNET_NSI>(k, L.N, s);
        return L.W <= 16 ? launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x7B0, false, UNET_NSI>(k, L.N, s) : launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x3B0, false, UNET_NSI>(k, L.N, s);
    }
    if (L.conv7v) {        // 7 x 1 column conv (padding 3 rows, zero or reflected) as three vertically displaced 3-tap blocks: panels from conv_pack7v, slab output
        if (!pc || L.out_mode != OUT_SLAB || (nt != 2 && nt != 4) || L.res1 || L.res2 || L.up || L.act > 2 || L.y0 != 0 || k.y1 != L.H || L.reflect == 2 ||
            (long)L.H * L.W * 64 >= 0x7fffffffL)
            return set_error(INNFER_ERR_UNSUPPORTED, "conv7x1: slab output of 32- / 64-channel tiles, no residual / upsampling, images below 33 M pixels");
        k.nchunks = 3 * k.ncg; k.s9v = 1;
        if (L.stats_part) return launch_pc<2, 4, 4, OUT_SLAB, true, false, 0x1092>(k, L.N, s);
        return nt == 4 ? launch_pc<2, 4, 4, OUT_SLAB, true, false, 0x92>(k, L.N, s) : launch_pc<3, 2, 4, OUT_SLAB, true, false, 0x92>(k, L.N, s);
    }
    if (L.conv1x1) {        // centre tap only: panels from conv_pack_1x1
        if (!pc || L.out_mode != OUT_SLAB || (nt != 2 && nt != 4))
            return set_error(INNFER_ERR_UNSUPPORTED, "conv1x1: slab outputs of 32- / 64-channel tiles on the producer-consumer kernel");
        if (L.prefix_lrelu) {
            if (nt != 4) return set_error(INNFER_ERR_UNSUPPORTED, "conv1x1: the running-sum operand is built for 64-channel tiles");
            return launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x810>(k, L.N, s);
        }
        // three input slots (a 1x1 stage has no halo): the loaders never pause between chunks -- about 1 % on PAN / PPON (kernel_experiments.txt 32)
        return nt == 4 ? launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x10, false, 3>(k, L.N, s) : launch_pc<2, 2, 4, OUT_SLAB, false, false, 0x10, false, 3>(k, L.N, s);
    }
    if (L.gate_w) {          // out = act(v * sigmoid(W v + b)), v = this conv's fp16 result (PAN's PA block behind an up-conv): conv3x3_pc<.., TMF | 0x80000>
        if (!pc || L.out_mode != OUT_SLAB || nt != 2 || L.K != 32 || L.res1 || L.res2 || L.act > 2 || L.reflect || L.stats_part)
            return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: the self gate belongs to 32-output slab convs without residuals (act = the activation AFTER the gate)");
        k.sg_w = L.gate_w; k.sg_bias = L.gate_bias;
        return launch_pc<3, 2, 4, OUT_SLAB, false, false, 0x801FF>(k, L.N, s);
    }
    if (pc && L.out_mode == OUT_SLAB && nt == 2 && !L.fuse_w) {
        // 32-output layers: 24-row tiles, two LDS stages.  pc 5 (diagnostic builds): 16-row tiles on the three-slot input ring (continuous LDS-DMA
        // issue) -- measured within +-1 % of the default on the frame and on the chop path (profiles/r2/kernel_experiments.txt), so the simpler form ships
        if (pc == 5) return launch_pc<2, 2, 4, OUT_SLAB, false, false, 0x1FF, false, 3>(k, L.N, s);
        if (INNFER_KNOB("INNFER_FAT", 0) & 1) return launch_pc<6, 2, 4, OUT_SLAB, false, false, 0x1FF, false, 2, 4>(k, L.N, s);
        return pc == 2 ? launch_pc<3, 2, 8>(k, L.N, s) : pc == 3 ? launch_pc<2, 2, 4>(k, L.N, s) : launch_pc<3, 2, 4>(k, L.N, s);
    }
    if (L.fuse_w) {          // HR_conv0 with the network's last conv in its epilogue (conv3x3_pc<.., TMF | 0x20000>) + the rim pass
        if (!conv_fuse_last_ok(L)) return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: the fused last conv needs 64 or 32 output channels, whole 16 x 32 tiles, act 0..2, no residual / upsampling / row range");
        k.fl_w = L.fuse_w; k.fl_bias = L.fuse_bias; k.fl_side = L.fuse_side; k.fl_out = L.fuse_out; k.fl_oc = L.fuse_oc; k.fl_out_mode = L.fuse_out_mode;
        if (nt == 2 && L.rowp) return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: the fused last conv behind a 32-channel conv takes lane-contiguous panels");
        if (int rc = nt == 2 ? launch_pc<2, 2, 4, OUT_SLAB, false, false, 0x201FF>(k, L.N, s)
                             : L.rowp ? launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x4201FF>(k, L.N, s) : launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x201FF>(k, L.N, s)) return rc;
        const long nthr = (long)L.N * (L.H / 16) * (L.W / 32) * 92;
        hipLaunchKernelGGL(fuse_combine_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, (const float*)L.fuse_side, L.fuse_bias, L.fuse_out, L.fuse_out_mode, L.out_denorm, L.out_round16, L.fuse_oc,
                           L.N, L.H, L.W);
        INNFER_HIP(hipGetLastError());
        return INNFER_OK;
    }
    if (L.out_mode == OUT_SHUFFLE2 && L.rowp == 2) {          // PixelShuffle(2) store on the producer / consumer kernel: phase-major plane-order panels (conv_pack_shuffle2)
        if (!pc || nt != 4 || L.K % 256 || L.res1 || L.res2 || L.up || L.reflect || L.act > 2 || L.y0 != 0 || k.y1 != L.H || L.stats_part ||
            (long)L.N * L.H * L.W * 64 * 4 >= 0x7fffffffL)
            return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: the phase-major PixelShuffle(2) store needs K %% 256 == 0 (64-channel phases), act 0..2, no residual / upsampling / row range, < 2 GiB per output group");
        k.phase_c = L.K / 4;
        return launch_pc<2, 4, 4, OUT_SLAB, false, false, 0xC001FF>(k, L.N, s);
    }
    if (pc && L.out_mode == OUT_SLAB && nt == 4 && L.stats_part) return launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x11FF>(k, L.N, s);
    if (pc && L.out_mode == OUT_SLAB && nt == 4 && L.res1_lds && L.res1 && L.res1 == L.in && L.res1_gstride == L.in_gstride && L.act == 0 && L.K == 64 && L.C >= 96 &&
        !L.up && !L.reflect && L.s1 != 0.f && (L.res1_lds == 2 || L.res2)) {
        // the dense block's last conv: the residual is the conv's own input groups 0 and 1 -- taken from the live LDS stages (conv3x3_pc<.., TMF | 0x40000>).
        // res1_lds 1 (the networks' default): only where a SECOND residual follows (the last dense block of an RRDB) -- there the epilogue could not batch two
        // residuals' loads beside the accumulators and ran 15 % behind the one-residual layers; with x from LDS it keeps ONE memory residual and batches it
        // (0.4096 -> 0.3945 ms per launch at 1080p).  The one-residual layers' batched loads were L2 hits already hidden behind the tile's last MFMAs: the LDS
        // form costs them +0.9 % (0.3574 -> 0.3607 ms), so they keep the epilogue load; res1_lds 2 forces the LDS form everywhere (profiles/r4/rlds_ab.txt).
        k.rs1 = 1.0f / L.s1;
        return L.rowp ? launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x4401FF>(k, L.N, s) : launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x401FF>(k, L.N, s);
    }
    if (pc && L.out_mode == OUT_SLAB && nt == 4) {
        // diagnostic builds: four consumer waves of twice the rows (measured within +-1 %: profiles/r2/kernel_experiments.txt 10)
        if (!L.rowp && (INNFER_KNOB("INNFER_FAT", 0) & 2)) return launch_pc<4, 4, 4, OUT_SLAB, false, false, 0x1FF, false, 2, 4>(k, L.N, s);
        return L.rowp ? launch_pc<2, 4, 4, OUT_SLAB, false, false, 0x4001FF>(k, L.N, s) : launch_pc<2, 4, 4>(k, L.N, s);
    }
    if (L.conv7) {         // 7x7 (padding 3) as nine displaced 3x3 convs: planar output, <= 16 output channels, panels from conv_pack7x7
        if (!pc || L.out_mode != OUT_NCHW || nt != 1 || L.res1 || L.res2 || L.up || (long)L.H * L.W * 64 >= 0x7fffffffL)
            return set_error(INNFER_ERR_UNSUPPORTED, "conv7x7: planar output with K <= 16, no residual / upsampling, images below 33 M pixels");
        k.nchunks = 9 * k.ncg;
        return launch_pc<3, 1, 4, OUT_NCHW, true>(k, L.N, s);
    }
    if (pc && L.out_mode == OUT_NCHW && nt == 1 && !L.res1 && !L.res2) return launch_pc<3, 1, 4, OUT_NCHW>(k, L.N, s);
    if (pc && L.out_mode == OUT_NCHW && nt == 2 && !L.res1 && !L.res2 && !L.phase_c && !L.reflect && !L.out_u8) return launch_pc<3, 2, 4, OUT_NCHW>(k, L.N, s);
    if (L.act >= 3 || L.phase_c > 0 || L.reflect)
        return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: tanh / phase output / gate epilogues / reflection padding exist only in the producer-consumer kernel");
    switch (L.out_mode) {
        case OUT_SLAB:
            if (nt == 4) return rpw64 == 4 ? launch_t<4, 4, OUT_SLAB>(k, L.N, s) : rpw64 == 3 ? launch_t<3, 4, OUT_SLAB>(k, L.N, s) : launch_t<2, 4, OUT_SLAB>(k, L.N, s);
            if (nt == 2) return rpw32 == 6 ? launch_t<6, 2, OUT_SLAB>(k, L.N, s) : rpw32 == 5 ? launch_t<5, 2, OUT_SLAB>(k, L.N, s) : rpw32 == 3 ? launch_t<3, 2, OUT_SLAB>(k, L.N, s) : launch_t<4, 2, OUT_SLAB>(k, L.N, s);
            return launch_t<4, 1, OUT_SLAB>(k, L.N, s);
        case OUT_NCHW:
            if (nt == 4) return launch_t<2, 4, OUT_NCHW>(k, L.N, s);
            if (nt == 2) return launch_t<4, 2, OUT_NCHW>(k, L.N, s);
            return launch_t<4, 1, OUT_NCHW>(k, L.N, s);
        case OUT_SHUFFLE2:
            if (nt == 4) return launch_t<2, 4, OUT_SHUFFLE2>(k, L.N, s);
            return set_error(INNFER_ERR_UNSUPPORTED, "pixelshuffle conv needs K %% 64 == 0");
    }
    return set_error(INNFER_ERR_INVALID, "conv3x3: bad out_mode");
}

}  // namespace innfer
