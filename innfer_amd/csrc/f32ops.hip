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
#include <atomic>
#include <algorithm>
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

// The LARGE-SHAPE FALLBACK of f32conv_launch (round 4's first form of the conv; kept on purpose): f32conv_tiled below addresses its patch with 32-bit byte offsets
// and bounded LDS tiles -- a launch whose input view spans 2 GiB or more (fp32 NCHW frames beyond ~8 M pixels x 64 channels), whose weight panel exceeds 2 GiB or whose best tile does not
// fit comes here: 64-bit addressing, operands straight from L1 / L2, ~0.12 of the fp32 matrix peak.  Every golden and bench shape takes the tiled kernel.
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
        const long opix0 = ((long)(oy[pt] * p.osy + p.ooy) * p.Wout + ox[pt] * p.osx + p.oox);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int k = k0 + 16 * t + 4 * lg + j;
                if (k >= p.K) continue;
                long opix = opix0;
                if (p.phase_k > 0) {                       // fused phases: channel k is phase ph of output channel k % phase_k
                    const int ph = k / p.phase_k;
                    k -= ph * p.phase_k;
                    opix = (long)(2 * oy[pt] + (ph >> 1)) * p.Wout + 2 * ox[pt] + (ph & 1);
                }
                float v = acc[t][pt][j] + (p.bias ? p.bias[k] : 0.f);
                if (p.mul) v = p.mul[(long)n * p.mul_nstride + (long)k * p.mul_cstride + opix] * (1.0f / (1.0f + expf(-v)));
                v = f32_act(v, p.act);
                if (p.oscale != 0.f) v *= p.oscale;
                if (p.res) v += p.res[(long)n * p.res_nstride + (long)k * p.res_cstride + opix];
                p.out[(long)n * p.out_nstride + (long)k * p.out_cstride + opix * p.out_pstride] = v;
            }
    }
}

// ---- LDS-tiled form (round 5) ------------------------------------------------------------------------------------------------
// The kernel above reads every MFMA operand with a scalar global load behind per-lane bounds tests (0.12 of the fp32 matrix peak).  Here a workgroup owns
// KT = 16 NKT output channels x 256 output pixels -- TY x TX pixels of IMG images, TY TX IMG = 256: 8 x 32 of one image down to 1 x 1 of 256 images, so the
// UNet's 8 x 8 .. 1 x 1 levels fill their pixel tiles with the batch -- and walks the input channels in chunks of CC:
//   patch    every source pixel a tap of the tile reads (PH x PW per channel and image) with the padding mode, the nearest-2x view and the input activation
//            resolved as it is written.  A thread's (at most NE) elements of a chunk have chunk-invariant offsets, derived once per tile; the NEXT chunk's values
//            are loaded into registers before the current chunk's MFMAs issue and written to LDS after them: HBM latency hides behind a whole compute phase
//            (the first version staged load -> wait -> write per chunk and ran at 1.0x of the direct kernel: latency-bound; profiles/r5/fp32_modes.txt);
//   weights  the chunk's rows [tap][c4][KT][4] by LDS-DMA (one 1-KB piece per row, range-checked zero fill beyond the panel) into the OTHER of two LDS
//            buffers, in flight during the same compute phase;
//   compute  wave w: its 4 pixel tiles (16 pixels each) x NKT channel tiles, per (tap, 4 channels) NKT + 4 ds_read_b32 for 4 NKT MFMAs, fragments of the
//            next step fetched before the current step's MFMAs issue.
// Same panels, same epilogue, same F32Conv views as the direct kernel; the sums run in another order (chunk-major instead of tap-major): fp32 round-off,
// inside the 1e-4 bound of the mode by two orders of magnitude.
struct F32Tile {
    int C4, Kp;                  // panel geometry (f32conv_pack)
    unsigned wbytes;             // panel bytes: the weight DMA's range check
    int dymin, dxmin;            // smallest tap displacement: patch row 0 / column 0 of tile (ty0, tx0) is source (ty0 * isy + dymin, tx0 * isx + dxmin)
    int PH, PW, plane, PS;       // patch rows, columns, floats per channel (IMG PH PW), floats between channel planes (padded: bank spread of the four k lanes)
    float rplane, rimg, rpw;     // reciprocals for the per-tile element decode
    int CC;                      // input channels per chunk (multiple of 4)
    int txs, tys, imgs;          // log2 TX, log2 TY, log2 IMG; TX TY IMG <= 256 (fewer: the last waves' pixel tiles are empty)
    int tiles_x, nkg;            // tile columns; channel groups of KT
    int vec4;                    // the epilogue moves four consecutive pixels per lane (aligned, unit-stride output rows: see the kernel's epilogue)
    int abl;                     // diagnostic build only (make ablate, INNFER_F32_ABL): skip 1 the MFMA steps, 2 the epilogue, 4 the patch loads, 8 the weight DMA
};

constexpr int F32_NE = 21;       // patch elements per thread and chunk: CC * plane <= 256 * NE

__device__ __forceinline__ int f32_fastdiv(int e, int d, float rd) {      // floor(e / d) for 0 <= e < 2^21 (rd = 1 / d): (e + 0.5) / d lies >= 0.5 / d from an integer
    (void)d;
    return (int)(((float)e + 0.5f) * rd);
}

// NPT: pixel tiles (16 pixels) per wave -- 4: a 256-pixel tile; 1: a 64-pixel tile for the grids whose tiles hold at most 64 live pixels (the UNet's <= 8 x 8
// levels): the same walk without the MFMAs of empty pixel tiles (a run-time skip of them cost the big layers 40 %: see below)
template <int NKT, int NPT>
__global__ __launch_bounds__(256) void f32conv_tiled(const F32Conv p, const F32Tile t) {
    extern __shared__ __attribute__((aligned(16))) float f32lds[];
    constexpr int NE = F32_NE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lg = lane >> 4;
    const int kg = blockIdx.x % t.nkg, tile = blockIdx.x / t.nkg;
    const int tyi = tile / t.tiles_x, txi = tile - tyi * t.tiles_x;
    const int TXm = (1 << t.txs) - 1, TYm = (1 << t.tys) - 1, IMG = 1 << t.imgs;
    const int oy0 = tyi << t.tys, ox0 = txi << t.txs, k0 = kg * 16 * NKT, n0 = blockIdx.z * IMG;
    const int CC4 = t.CC >> 2, nstep = p.ntap * CC4;
    float* patch = f32lds;
    char* wl0 = (char*)(f32lds + 2L * t.CC * t.PS);          // (two patch buffers in front)
    constexpr int RPI = NKT == 3 ? 1 : 4 / NKT;                  // weight rows per 1-KB LDS-DMA piece: a row is 16 NKT channels x 16 bytes (NKT 3: 768 of a 1-KB pitch)
    constexpr int ROWB = 1024 / RPI;                            // bytes between rows in LDS
    const long wl_bytes = (long)((nstep + RPI - 1) / RPI) * 1024;
    [[maybe_unused]] const float* inb = p.in + (long)n0 * p.in_nstride;
    const int Hv = p.up ? 2 * p.Hin : p.Hin, Wv = p.up ? 2 * p.Win : p.Win;
    const int img_px = t.PH * t.PW;
    // ---- this thread's patch elements e = tid + 256 i of a chunk: global offset from (image n0, channel c0) or -1 (padding, beyond the batch), LDS offset
    const int nfull = (t.CC * t.plane) >> 8, nrem = (t.CC * t.plane) & 255;      // whole 256-thread slots of a chunk's elements, threads of the partial one
    int goff[NE], loff[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = tid + 256 * i;
        const int cl = f32_fastdiv(e, t.plane, t.rplane), rem = e - cl * t.plane;
        const int il = f32_fastdiv(rem, img_px, t.rimg), r2 = rem - il * img_px;
        const int y = f32_fastdiv(r2, t.PW, t.rpw), x = r2 - y * t.PW;
        int vy = oy0 * p.isy + t.dymin + y, vx = ox0 * p.isx + t.dxmin + x;
        if (p.pad_mode == 1) {
            vy = vy < 0 ? -vy : (vy >= Hv ? 2 * Hv - 2 - vy : vy);
            vx = vx < 0 ? -vx : (vx >= Wv ? 2 * Wv - 2 - vx : vx);
        } else if (p.pad_mode == 2) {
            vy = min(max(vy, 0), Hv - 1); vx = min(max(vx, 0), Wv - 1);
        }
        const bool in_chunk = cl < t.CC;
        const bool ok = in_chunk && il < IMG && n0 + il < p.N && vy >= 0 && vy < Hv && vx >= 0 && vx < Wv;
        const int iy = p.up ? vy >> 1 : vy, ix = p.up ? vx >> 1 : vx;
        goff[i] = ok ? ((int)(cl * p.in_cstride + il * p.in_nstride) + iy * p.Win + ix) * 4 : (int)0x80000000;
        loff[i] = in_chunk ? cl * t.PS + rem : 0;                  // (slots past the chunk's elements are never loaded / written: nfull, nrem below)
    }
    // ---- B fragment bases of the wave's four pixel tiles (floats into the patch, without tap and chunk-channel terms)
    int boff[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        const int q = (pt * 4 + wave) * 16 + li;                  // pixel tiles are dealt to the waves round-robin: a tile of fewer than 256 pixels still feeds all four
        const int x = q & TXm, y = (q >> t.txs) & TYm, il = min(q >> (t.txs + t.tys), IMG - 1);      // (pixels beyond the tile's images: computed on image IMG - 1, never stored)
        boff[pt] = il * img_px + y * p.isy * t.PW + x * p.isx + lg * t.PS;
    }
    // (pixel tiles without a live pixel still run their MFMAs: a wave-uniform skip made hipcc shuffle the 64 accumulator registers through copies on every
    //  step -- 48 v_accvgpr_mov per 16 MFMAs, big layers 84 -> 50 TFLOP/s -- and the layers with such tiles wait for their weight stream, not for the pipe)
    // tap -> patch displacement, lane `tap` of one register (read with v_readlane per step: indexing the kernel-argument arrays per step is a scalar memory
    // load whose latency a step of few MFMAs cannot hide -- the 1 x 1 levels ran 250 cycles per 32-cycle step on it)
    const int tapv = lane < p.ntap ? (p.dy[lane] - t.dymin) * t.PW + (p.dx[lane] - t.dxmin) : 0;
    f32x4 acc[NKT][NPT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) acc[kt][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

    float v[NE];
    // (buffer loads: one instruction per element -- the per-element 64-bit address arithmetic of plain loads was a third of a chunk's ~600 fixed instructions, and the
    //  chunk's fixed cost is what the UNet's <= 8 x 8 levels pay 30 .. 40 times per workgroup: profiles/r5/f32_ablate_deep.txt.  goff holds BYTE offsets; an element
    //  outside the image / batch carries the out-of-range offset and loads zero; the descriptor is rebuilt per chunk on the chunk's first channel)
    auto prefetch_patch = [&](int c0) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef INNFER_ABLATE
        if (t.abl & 4) {
#pragma unroll
            for (int i = 0; i < NE; ++i) v[i] = 0.f;
            return;
        }
#endif
        const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)(inb + (long)c0 * p.in_cstride), 0, 0x7fffffff, 0x00020000);
        if (c0 + t.CC <= p.C) {
#pragma unroll
            for (int i = 0; i < NE; ++i)
                if (i <= nfull) v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, goff[i], 0, 0));          // (slot nfull: its tail threads carry the out-of-range offset)
        } else {                                                  // the last chunk of a channel count that is no multiple of CC: channels beyond C read zero
            const int lim = (p.C - c0) * t.plane;
#pragma unroll
            for (int i = 0; i < NE; ++i)
                if (i <= nfull) v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, tid + 256 * i < lim ? goff[i] : (int)0x80000000, 0, 0));
        }
#else
        (void)c0;
#endif
    };
    auto issue_weights = [&](int c0, int buf) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef INNFER_ABLATE
        if (t.abl & 8) return;
#endif
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, t.wbytes, 0x00020000);
        char* dst = wl0 + buf * wl_bytes;
        const int c40 = c0 >> 2;
        // piece j (1 KB) holds rows j RPI .. j RPI + RPI - 1: lane -> (its row of the piece, channel of the row); wave w takes pieces w, w + 4, ..
        constexpr int LPR = 64 / RPI;                            // lanes per row
        const int sub = lane / LPR, kk = lane - sub * LPR;
        const bool lane_ok = kk < 16 * NKT && k0 + kk < t.Kp;
        const int npieces = (nstep + RPI - 1) / RPI;
        int r = wave * RPI + sub;                                // this lane's row, stepped by 4 RPI rows per piece without divisions
        int tap = r / CC4, c4l = r - tap * CC4;
        const int dtap = (4 * RPI) / CC4, dc4 = 4 * RPI - dtap * CC4;
        for (int j = wave; j < npieces; j += 4, r += 4 * RPI, tap += dtap, c4l += dc4) {
            if (c4l >= CC4) { c4l -= CC4; ++tap; }
            const unsigned row = (unsigned)(((tap * t.C4 + c40 + c4l) * t.Kp + k0) * 16);
            const int voff = (lane_ok && r < nstep && c40 + c4l < t.C4) ? (int)(row + kk * 16) : (int)0x80000000;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, voff, 0, 0, 0);
        }
#else
        (void)c0; (void)buf;
#endif
    };

    // The patch is double-buffered in LDS and its values run TWO chunks ahead in registers: chunk i's steps read patch buffer i & 1 while the writes of chunk i + 1
    // (into the other buffer) and the loads of chunk i + 2 (into registers) are in flight behind them -- one barrier per chunk, at its end.  (With one patch buffer
    // a chunk was wait -> barrier -> 21 LDS writes -> barrier -> steps: the writes and both barriers stood between the MFMA phases, and the two workgroups of a CU did
    // not fill each other's gaps: matrix pipe 0.55 busy, profiles/r5/pmc_unet_fp32.txt.)
    auto write_patch = [&](int pb) __attribute__((always_inline)) {
        float* pw = patch + (long)pb * t.CC * t.PS;
        // (slots 0 .. nfull - 1 hold an element for every thread, slot nfull for the first nrem threads, later slots none: uniform tests, no per-element predicate)
#define INNFER_F32_WRITE(EXPR)                                                                        \
        _Pragma("unroll") for (int i = 0; i < NE; ++i) {                                              \
            if (i < nfull || (i == nfull && tid < nrem)) pw[loff[i]] = (EXPR);                        \
        }
        if (p.in_act == 1) { INNFER_F32_WRITE(v[i] > 0.f ? v[i] : 0.2f * v[i]) }
        else if (p.in_act == 2) { INNFER_F32_WRITE(fmaxf(v[i], 0.f)) }
        else { INNFER_F32_WRITE(v[i]) }
#undef INNFER_F32_WRITE
    };
    prefetch_patch(0);
    issue_weights(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    write_patch(0);
    if (t.CC < p.C) prefetch_patch(t.CC);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int c0 = 0; c0 < p.C; c0 += t.CC, buf ^= 1) {
        if (c0 + t.CC < p.C) {
            write_patch(buf ^ 1);                                  // chunk i + 1's values: in registers since the wait in front of the last barrier
            issue_weights(c0 + t.CC, buf ^ 1);
            if (c0 + 2 * t.CC < p.C) prefetch_patch(c0 + 2 * t.CC);
        }
        const float* pcur = patch + (long)buf * t.CC * t.PS;
        // ---- compute: (tap, 4 channels) steps, the next step's fragments in flight behind the current MFMAs
        const float* wl = (const float*)(wl0 + buf * wl_bytes);
        // G steps per group: a step of NKT NPT MFMAs keeps the pipe busy for 32 NKT NPT cycles; the narrow forms fetch several steps' fragments at a time.  Two groups
        // per trip on alternating register sets.  (Hand-counted waits around inline-asm reads were tried -- hipcc waits for part of the NEXT group's reads in front of
        // the CURRENT group's MFMAs -- and changed nothing: the matrix pipe of these kernels is 0.55 busy for other reasons, profiles/r5/fp32_modes.txt.)
        constexpr int G = NKT * NPT >= 8 ? 1 : 4;
        float fa0[G][NKT], fb0[G][NPT], fa1[G][NKT], fb1[G][NPT];
        int ftap = 0, fc4 = 0;
        auto fetch_group = [&](float (&fa)[G][NKT], float (&fb)[G][NPT]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < G; ++u) {
                if (ftap < p.ntap) {
                    const int toff = __builtin_amdgcn_readlane(tapv, ftap) + fc4 * 4 * t.PS;
                    const float* wa = wl + (long)(ftap * CC4 + fc4) * (ROWB / 4) + li * 4 + lg;
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) fa[u][kt] = wa[kt * 64];
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt) fb[u][pt] = pcur[boff[pt] + toff];
                    if (++fc4 == CC4) { fc4 = 0; ++ftap; }
                } else {
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) fa[u][kt] = 0.f;
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt) fb[u][pt] = 0.f;
                }
            }
        };
        auto mfma_group = [&](const float (&fa)[G][NKT], const float (&fb)[G][NPT]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < G; ++u)
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) acc[kt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][kt], fb[u][pt], acc[kt][pt], 0, 0, 0);
        };
#ifdef INNFER_ABLATE
        const int ngroups = (t.abl & 1) ? 0 : (nstep + G - 1) / G;
#else
        const int ngroups = (nstep + G - 1) / G;
#endif
        fetch_group(fa0, fb0);
        for (int g = 0; g < ngroups; g += 2) {
            fetch_group(fa1, fb1);                                 // group g + 1 (zeros beyond the last step: its MFMAs add nothing)
            mfma_group(fa0, fb0);
            if (g + 2 < ngroups) fetch_group(fa0, fb0);
            mfma_group(fa1, fb1);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); // the next chunk's patch writes and weight pieces (issued in front of this chunk's steps) are in LDS
        __syncthreads();                                           // ... every wave's, and every wave has read this chunk's fragments
    }
    // ---- epilogue: bias, gate, activation, scale, residual, store through the output view -- the direct kernel's arithmetic per value, ROW-WISE: the MFMA
    // result layout gives a lane 4 channels of ONE pixel, so a store instruction of the value-by-value form wrote four 64-byte pieces of four channel planes (and
    // its gate / residual loads read such pieces); at 20 .. 40 channels that epilogue was 21 % (3 x 3) to 61 % (1 x 1) of a PAN conv (profiles/r5/f32_ablate_pan.txt).
    // The tile goes through LDS once ([channel][pixel], pitch TP + 4: conflict-free both ways), then wave w takes channels w, w + 4, ..: with the output rows
    // 16-byte aligned a lane moves FOUR consecutive pixels -- one b128 LDS read, one 16-byte store per channel and 256 pixels -- else one pixel per lane, still whole
    // 256-byte runs of a plane; the bias is a scalar per row.
#ifdef INNFER_ABLATE
    if (t.abl & 2) return;
#endif
    constexpr int TP = 64 * NPT, QP = TP + 4;
    __syncthreads();                                               // the last chunk's fragments have been read: the LDS is free
    float* ot = f32lds;
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) ot[(16 * kt + 4 * lg + j) * QP + (pt * 4 + wave) * 16 + li] = acc[kt][pt][j];
    __syncthreads();
    auto finish = [&](float f, int k, long nbase_mul, long nbase_res, long opix) __attribute__((always_inline)) {
        if (p.mul) f = p.mul[nbase_mul + (long)k * p.mul_cstride + opix] * (1.0f / (1.0f + expf(-f)));
        f = f32_act(f, p.act);
        if (p.oscale != 0.f) f *= p.oscale;
        if (p.res) f += p.res[nbase_res + (long)k * p.res_cstride + opix];
        return f;
    };
    if (t.vec4) {
        // four consecutive pixels of a row per lane (host: osx = 1, pixel stride 1, no fused phases, Wo / Wout / the view strides multiples of 4, 16-byte aligned bases)
        const int q = 4 * lane;
        if (q < TP) {
            const int ox = ox0 + (q & TXm), oy = oy0 + ((q >> t.txs) & TYm), il = q >> (t.txs + t.tys), n = n0 + il;
            if (oy < p.Ho && ox < p.Wo && il < IMG && n < p.N) {
                const long opix = (long)(oy * p.osy + p.ooy) * p.Wout + ox + p.oox;
                for (int kl = wave; kl < 16 * NKT; kl += 4) {
                    const int k = k0 + kl;
                    if (k >= p.K) break;
                    f32x4 v = *(const f32x4*)(ot + kl * QP + q);
                    const float bk = p.bias ? p.bias[k] : 0.f;
                    f32x4 m4 = f32x4{0.f, 0.f, 0.f, 0.f}, r4 = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (p.mul) m4 = *(const f32x4*)(p.mul + (long)n * p.mul_nstride + (long)k * p.mul_cstride + opix);
                    if (p.res) r4 = *(const f32x4*)(p.res + (long)n * p.res_nstride + (long)k * p.res_cstride + opix);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float f = v[e] + bk;
                        if (p.mul) f = m4[e] * (1.0f / (1.0f + expf(-f)));
                        f = f32_act(f, p.act);
                        if (p.oscale != 0.f) f *= p.oscale;
                        if (p.res) f += r4[e];
                        v[e] = f;
                    }
                    *(f32x4*)(p.out + (long)n * p.out_nstride + (long)k * p.out_cstride + opix) = v;
                }
            }
        }
        return;
    }
    for (int kl = wave; kl < 16 * NKT; kl += 4) {
        int k = k0 + kl;
        if (k >= p.K) break;
        int ph = 0;
        if (p.phase_k > 0) { ph = k / p.phase_k; k -= ph * p.phase_k; }          // fused phases: channel k0 + kl is phase ph of output channel k
        const float bk = p.bias ? p.bias[k] : 0.f;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int q = lane + 64 * i;
            const int ox = ox0 + (q & TXm), oy = oy0 + ((q >> t.txs) & TYm), il = q >> (t.txs + t.tys), n = n0 + il;
            if (oy >= p.Ho || ox >= p.Wo || il >= IMG || n >= p.N) continue;
            const long opix = p.phase_k > 0 ? (long)(2 * oy + (ph >> 1)) * p.Wout + 2 * ox + (ph & 1) : (long)(oy * p.osy + p.ooy) * p.Wout + ox * p.osx + p.oox;
            const float f = finish(ot[kl * QP + q] + bk, k, (long)n * p.mul_nstride, (long)n * p.res_nstride, opix);
            p.out[(long)n * p.out_nstride + (long)k * p.out_cstride + opix * p.out_pstride] = f;
        }
    }
}

template <int NKT, int NPT>
static int f32conv_tiled_launch(const F32Conv& k, const F32Tile& t, int tiles, int zgroups, size_t lds, hipStream_t s) {
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        INNFER_HIP(hipFuncSetAttribute((const void*)f32conv_tiled<NKT, NPT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done.fetch_or(bit, std::memory_order_release);
    }
#ifdef INNFER_ABLATE
    if (getenv("INNFER_F32_OCC")) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)f32conv_tiled<NKT, NPT>, 256, lds);
        fprintf(stderr, "[f32conv_tiled<%d,%d>] grid %d x %d lds %zu CC %d IMG %d -> %d workgroups / CU\n", NKT, NPT, tiles * t.nkg, zgroups, lds, t.CC, 1 << t.imgs, nb);
    }
#endif
    hipLaunchKernelGGL((f32conv_tiled<NKT, NPT>), dim3((unsigned)(tiles * t.nkg), 1, (unsigned)zgroups), dim3(256), lds, s, k, t);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
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

// The same with the plane held in registers (HW <= 256 VPT: every plane of the networks up to 128 x 128): ONE read of the plane instead of three.  Thread tid holds
// elements tid, tid + 256, .. exactly as the loops above walk them, and the block reductions are the same trees: the results are the same bits as f32_norm_kernel's.
template <int VPT>
__global__ __launch_bounds__(256) void f32_norm_reg_kernel(const float* in, long in_ns, long in_cs, float* out, long out_ns, long out_cs, int C, int HW, int mode, float eps,
                                                           const float* weight, const float* bias, int act, const float* res, long res_ns, long res_cs) {
    __shared__ float red[256];
    const int c = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const float* x = in + (long)n * in_ns + (long)c * in_cs;
    float v[VPT];
#pragma unroll
    for (int j = 0; j < VPT; ++j) v[j] = tid + 256 * j < HW ? x[tid + 256 * j] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) if (tid + 256 * j < HW) s += v[j];
    red[tid] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (tid < w) red[tid] += red[tid + w]; __syncthreads(); }
    const float mean = red[0] / (float)HW;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) if (tid + 256 * j < HW) { const float d = v[j] - mean; q += d * d; }
    red[tid] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (tid < w) red[tid] += red[tid + w]; __syncthreads(); }
    const float var = red[0] / (float)HW;
    const float inv = 1.0f / sqrtf(var + eps);
    const float al = mode == 2 ? inv : inv * weight[c], sh = mode == 2 ? -mean * inv : bias[c] - mean * inv * weight[c];
    float* y = out + (long)n * out_ns + (long)c * out_cs;
    const float* rr = res ? res + (long)n * res_ns + (long)c * res_cs : nullptr;
#pragma unroll
    for (int j = 0; j < VPT; ++j)
        if (tid + 256 * j < HW) y[tid + 256 * j] = f32_act(v[j] * al + sh, act) + (rr ? rr[tid + 256 * j] : 0.f);
}

// Planes of at most 64 values (the <= 8 x 8 levels): one WAVE per plane, four planes per block -- a 256-thread block with two block reductions per 1 .. 64 values was
// 46 us per launch whatever the size (eight such launches per UNet forward).  Sums over the lanes by xor butterflies (a fixed order; not the block tree's: results
// differ from f32_norm_kernel's in the last bits, not between a batch and its images).
__global__ __launch_bounds__(256) void f32_norm_small_kernel(const float* in, long in_ns, long in_cs, float* out, long out_ns, long out_cs, int C, int HW, int N, int mode, float eps,
                                                             const float* weight, const float* bias, int act, const float* res, long res_ns, long res_cs) {
    const int lane = threadIdx.x & 63, plane = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= C * N) return;
    const int n = plane / C, c = plane - n * C;
    const float* x = in + (long)n * in_ns + (long)c * in_cs;
    const float v = lane < HW ? x[lane] : 0.f;
    float s = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)HW;
    const float d = lane < HW ? v - mean : 0.f;
    float q = d * d;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float var = q / (float)HW;
    const float inv = 1.0f / sqrtf(var + eps);
    const float al = mode == 2 ? inv : inv * weight[c], sh = mode == 2 ? -mean * inv : bias[c] - mean * inv * weight[c];
    if (lane < HW) {
        const float r = res ? res[(long)n * res_ns + (long)c * res_cs + lane] : 0.f;
        out[(long)n * out_ns + (long)c * out_cs + lane] = f32_act(v * al + sh, act) + r;
    }
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

// The same for W == 4 wp (H / 4 x W / 4 pooling: every PAN frame whose width is a multiple of 4): a thread owns the strip X0 = 4 k + 2 .. 4 k + 5 of one row and channel,
// whose four pixels share floor(rx) == k, i.e. the same 4 x 4 taps -- 16 gathered reads for four outputs instead of 64 (the one-pixel kernel was 0.27 ms of PAN's 5.1 ms
// fp32 forward at 540 x 960, TA-bound).  The same products in the same order per output: identical bits.
// att here is CHANNEL-major ([N][C][hp * wp], f32_att_transpose_kernel below): the strips of a wave are consecutive k, so a tap's 64 reads are consecutive floats; with the
// attention kernel's pixel-major rows they were 64 cache lines (160-byte stride) and the kernel sat at 0.76 TB/s.
__global__ __launch_bounds__(256) void f32_fsa_combine_x4_kernel(const float* att, int hp, int wp, int C, const float* inp, float* out, int N, int H, int W, const float* gamma) {
    const int ns = wp + 1;                                   // strips k = -1 .. wp - 1 of a row
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)N * C * H * ns) return;
    const int k = (int)(t % ns) - 1, Y = (int)((t / ns) % H), c = (int)((t / ((long)ns * H)) % C);
    const long n = t / ((long)ns * H * C);
    const int X0 = 4 * k + 2;
    const float A = -0.75f;
    const float sy = (float)hp / (float)H, sx = (float)wp / (float)W;
    const float ry = sy * ((float)Y + 0.5f) - 0.5f;
    const int iy = (int)floorf(ry);
    const float ty = ry - (float)iy;
    const float wy[4] = {cub2(ty + 1.f, A), cub1(ty, A), cub1(1.f - ty, A), cub2(2.f - ty, A)};
    float wx[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float rx = sx * ((float)(X0 + j) + 0.5f) - 0.5f;
        const float tx = rx - (float)k;                      // floor(rx) == k for the strip's four pixels
        wx[j][0] = cub2(tx + 1.f, A); wx[j][1] = cub1(tx, A); wx[j][2] = cub1(1.f - tx, A); wx[j][3] = cub2(2.f - tx, A);
    }
    const float* an = att + (n * C + c) * (long)hp * wp;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), hp - 1);
        float q[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) q[b] = an[(long)yy * wp + min(max(k - 1 + b, 0), wp - 1)];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float row = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) row += q[b] * wx[j][b];
            v[j] += row * wy[a];
        }
    }
    const float gm = gamma[0];
    const long o = ((n * C + c) * H + Y) * (long)W + X0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (X0 + j >= 0 && X0 + j < W) out[o + j] = gm * v[j] + inp[o + j];
}

// [N][Np][C] -> [N][C][Np] (the attention's rows, 5 MB at 540 x 960) through a 32 x 33 LDS tile
__global__ __launch_bounds__(256) void f32_att_transpose_kernel(const float* att, float* att_t, int Np, int C) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (p0 + r < Np && c0 + tx < C) tile[r][tx] = att[((long)n * Np + p0 + r) * C + c0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (c0 + r < C && p0 + tx < Np) att_t[((long)n * C + c0 + r) * Np + p0 + tx] = tile[tx][r];
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
    if (L.phase_k && (L.phase_k < 0 || L.K != 4 * L.phase_k || L.osy != 1 || L.osx != 1 || L.ooy || L.oox || (L.out_pstride > 1)))
        return set_error(INNFER_ERR_INVALID, "f32conv: fused phases take K = 4 * phase_k channels on the input grid (osy = osx = 1, no offset)");
    const int C4 = (L.C + 3) / 4, Kp = (L.K + 31) / 32 * 32, nkt = Kp / 32;
    const long npx = (long)L.Ho * L.Wo;
    F32Conv k = L;
    if (k.out_pstride == 0) k.out_pstride = 1;
    GtScope gt(s, "f32conv (fp32 MFMA, -no_fp16 mode)", 2.0 * L.N * (double)npx * L.K * L.C * L.ntap, (double)L.N * npx * (L.C * L.ntap / (double)(L.isy * L.isx) + L.K) * 4.0);
#ifdef INNFER_ABLATE
    static const int f32_direct = getenv("INNFER_F32_DIRECT") ? atoi(getenv("INNFER_F32_DIRECT")) : 0;      // A/B: the direct (round-4) kernel for every shape
#else
    constexpr int f32_direct = 0;
#endif
    // ---- the LDS-tiled kernel: 256-pixel tiles of 8 x 32 / 16 x 16 pixels of one image, or of the whole (<= 8 x 8) grids of several images ----
    if (!f32_direct) {
        F32Tile t{};
        t.C4 = C4; t.Kp = Kp;
        int dymin = L.dy[0], dymax = L.dy[0], dxmin = L.dx[0], dxmax = L.dx[0];
        for (int i = 1; i < L.ntap; ++i) {
            dymin = std::min(dymin, L.dy[i]); dymax = std::max(dymax, L.dy[i]);
            dxmin = std::min(dxmin, L.dx[i]); dxmax = std::max(dxmax, L.dx[i]);
        }
        auto clog2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
        t.txs = std::min(L.Wo > 16 ? 5 : 4, clog2(L.Wo));
        t.tys = std::min(8 - t.txs, clog2(L.Ho));
        const int TX = 1 << t.txs, TY = 1 << t.tys;
        t.dymin = dymin; t.dxmin = dxmin;
        t.PH = (TY - 1) * L.isy + (dymax - dymin) + 1;
        t.PW = (TX - 1) * L.isx + (dxmax - dxmin) + 1;
        // Tile shape by a small cost model instead of rules (round 5: the rules left the UNet's 8 x 8 levels on half-empty 256-pixel tiles and its 4 x 4 .. 1 x 1
        // levels with 4-channel chunks -- 6.0 of 15.4 ms for 18 % of the FLOPs).  Candidates: pixel tiles per wave NPT in {1, 4} (64- / 256-pixel tiles), images per
        // tile IMG (powers of two that fit the tile), channels per chunk CC (multiples of 4 whose patch fits a thread's NE elements and the LDS), channel tiles NKT.
        // Estimated cycles of a launch = rounds x max(one workgroup's life, the matrix-pipe time of the workgroups that share a CU):
        //   life = setup + chunks x (steps x (NKT x NPT x 32 + (NKT + NPT) x 16) + per-chunk overhead),  rounds = ceil(workgroups / (256 CUs x co-resident workgroups)).
        // (IMG, NPT, CC) are chosen for a NOMINAL batch of 64 -- never from the real one: CC is the order of the sums, and a batch must equal its images' own
        // forwards bit for bit; NKT (which does not touch the sums) is then chosen for the real batch.
        const int cmax = (L.C + 3) / 4 * 4, nkt16 = (L.K + 15) / 16;          // nkt16: 16-channel tiles that hold real outputs
        const int px1 = t.PH * t.PW, tile1 = 1 << (t.txs + t.tys);
        const int tiles_y = (L.Ho + TY - 1) / TY;
        t.tiles_x = (L.Wo + TX - 1) / TX;
        const int tiles = t.tiles_x * tiles_y;
        auto lds_need = [&](int c, int img, int nktc, int npt) {
            const size_t plane = (size_t)img * px1, ps = plane + 80;
            const size_t rpi = nktc == 3 ? 1 : 4 / nktc, wb = ((size_t)L.ntap * (c / 4) + rpi - 1) / rpi * 1024;      // weight rows are 256 NKT bytes apart (four / two per DMA piece)
            return std::max(2 * (size_t)c * ps * 4 + 2 * wb, (size_t)16 * nktc * (64 * npt + 4) * 4);
        };
#ifdef INNFER_ABLATE
        static const double unhidden = getenv("INNFER_F32_UNHIDDEN") ? atof(getenv("INNFER_F32_UNHIDDEN")) : 0.0;      // A/B of the model's constants (diagnostic build)
        static const size_t lds_cap = (getenv("INNFER_F32_LDSCAP") ? atoi(getenv("INNFER_F32_LDSCAP")) : 80) * 1024;
#else
        constexpr double unhidden = 0.0;           // (a term for the part of a chunk's fixed cost the co-resident workgroup does not hide: 1200 cycles bought the UNet 1 % and cost PAN 7.5 %, scripts/r5/f32_model_ab.sh)
        constexpr size_t lds_cap = 80 * 1024;
#endif
        auto est = [&](int npt, int img, int c, int nktc, int batch) -> double {
            const size_t lds = lds_need(c, img, nktc, npt);
            const int co_max = (int)std::min<size_t>(2, (160 * 1024) / lds);                    // (two waves per SIMD by registers)
            if (co_max < 1) return 1e30;
            const long wgs = (long)tiles * ((batch + img - 1) / img) * ((nkt16 + nktc - 1) / nktc);
            const int chunks = (cmax + c - 1) / c;
            const double mfma = (double)L.ntap * (c / 4) * (nktc * npt * 32.0 + (nktc + npt) * 16.0);      // (+ ~16 cycles per fragment read that the MFMAs do not cover: favours wide steps -- the b32-fed loop tops out at 0.76 of the pipe, profiles/r5/mfma_f32_micro.txt)
            const double life = 6000.0 + chunks * (mfma + 2500.0);
            const long slots = 256L * co_max;
            const long rounds = (wgs + slots - 1) / slots;
            const int co = (int)std::min<long>(co_max, (wgs + 255) / 256);
            return (double)rounds * std::max(life, (double)co * chunks * (mfma + unhidden));
        };
        int b_npt = 4, b_imgs = 0, cc = 4;
        {
            double best = 1e30;
            for (int npt = 1; npt <= 4; npt += 3) {
                if (tile1 > 64 * npt) continue;
                for (int imgs = 0; (tile1 << imgs) <= 64 * npt; ++imgs) {
                    for (int c = std::min(128, cmax); c >= 4; c -= 4) {
                        if ((long)c * (px1 << imgs) > 256L * F32_NE) continue;
                        double e = 1e30;
                        for (int nktc = std::min(4, nkt16); nktc >= 1; nktc = nktc == 3 ? 2 : nktc >> 1) {
                            if (lds_need(c, 1 << imgs, nktc, npt) > lds_cap && !(c == 4 && imgs == 0)) continue;
                            e = std::min(e, est(npt, 1 << imgs, c, nktc, 64));
                        }
                        if (e < best * 0.999) { best = e; b_npt = npt; b_imgs = imgs; cc = c; }      // (every CC: a smaller one may admit a wider NKT within the LDS)
                    }
                }
            }
        }
        t.imgs = b_imgs;
        const int IMG = 1 << t.imgs;
        const bool npt1 = b_npt == 1;
        t.plane = IMG * px1;
        t.PS = t.plane + ((16 - t.plane % 64) + 64) % 64;          // plane stride = 16 (mod 64 banks): the four k lanes of a fragment read land 16 banks apart
        t.rplane = 1.0f / (float)t.plane; t.rimg = 1.0f / (float)px1; t.rpw = 1.0f / (float)t.PW;
        const int zgroups = (L.N + IMG - 1) / IMG;
        int NKT = std::min(4, nkt16);
        {
            double best = 1e30;
            for (int nktc = std::min(4, nkt16); nktc >= 1; nktc = nktc == 3 ? 2 : nktc >> 1) {
                const double e = est(b_npt, IMG, cc, nktc, L.N);
                if (e < best * 0.999) { best = e; NKT = nktc; }
            }
        }
        t.nkg = (nkt16 + NKT - 1) / NKT;
        auto lds_total = [&](int c) {                                 // two patch buffers, two weight buffers
            const size_t rpi = NKT == 3 ? 1 : 4 / NKT;
            return 2 * (size_t)c * t.PS * 4 + 2 * (((size_t)L.ntap * (c / 4) + rpi - 1) / rpi * 1024);
        };
        const size_t wbytes = f32conv_packed_floats(L.K, L.C, L.ntap) * 4;
        if ((long)cc * t.plane <= 256L * F32_NE && lds_total(cc) <= 150 * 1024 && wbytes < 0x7fffffffu &&
            ((long)cc * L.in_cstride + (long)IMG * L.in_nstride + (long)L.Hin * L.Win) * 4 < 0x7fffffffL && (long)cc * t.plane < (1 << 21)) {
            t.CC = cc; t.wbytes = (unsigned)wbytes;
            auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
            t.vec4 = TX >= 4 && L.osx == 1 && k.out_pstride == 1 && !L.phase_k && L.Wo % 4 == 0 && L.Wout % 4 == 0 && L.oox % 4 == 0 &&
                     L.out_nstride % 4 == 0 && L.out_cstride % 4 == 0 && al16(L.out) &&
                     (!L.res || (L.res_nstride % 4 == 0 && L.res_cstride % 4 == 0 && al16(L.res))) && (!L.mul || (L.mul_nstride % 4 == 0 && L.mul_cstride % 4 == 0 && al16(L.mul)));
#ifdef INNFER_ABLATE
            t.abl = getenv("INNFER_F32_ABL") ? atoi(getenv("INNFER_F32_ABL")) : 0;
#endif
            const size_t lds = std::max(lds_total(cc), (size_t)16 * NKT * ((npt1 ? 64 : 256) + 4) * 4);      // (>= the epilogue's [channel][pixel] tile)
            switch (NKT * 2 + (npt1 ? 1 : 0)) {
                case 2: return f32conv_tiled_launch<1, 4>(k, t, tiles, zgroups, lds, s);
                case 3: return f32conv_tiled_launch<1, 1>(k, t, tiles, zgroups, lds, s);
                case 4: return f32conv_tiled_launch<2, 4>(k, t, tiles, zgroups, lds, s);
                case 5: return f32conv_tiled_launch<2, 1>(k, t, tiles, zgroups, lds, s);
                case 6: return f32conv_tiled_launch<3, 4>(k, t, tiles, zgroups, lds, s);
                case 7: return f32conv_tiled_launch<3, 1>(k, t, tiles, zgroups, lds, s);
                case 8: return f32conv_tiled_launch<4, 4>(k, t, tiles, zgroups, lds, s);
                default: return f32conv_tiled_launch<4, 1>(k, t, tiles, zgroups, lds, s);
            }
        }
    }
    const int ktb = nkt >= 4 ? 4 : (nkt >= 2 ? 2 : 1);
    const long ptiles = (npx + 63) / 64;
    const int ptb = 4 / ktb;
    hipLaunchKernelGGL(f32conv_kernel, dim3((unsigned)((ptiles + ptb - 1) / ptb), (unsigned)((nkt + ktb - 1) / ktb), (unsigned)L.N), dim3(256), 0, s, k, ktb, C4, Kp);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

int f32_norm_launch(const float* in, long in_ns, long in_cs, float* out, long out_ns, long out_cs, int N, int C, long HW, int mode, float eps,
                    const float* weight, const float* bias, const float* rmean, const float* rvar, int act, hipStream_t s,
                    const float* res, long res_ns, long res_cs) {
    if ((mode == 0 || mode == 2) && HW <= 64) {                       // statistics of tiny planes: a wave per plane
        hipLaunchKernelGGL(f32_norm_small_kernel, dim3((unsigned)(((long)C * N + 3) / 4)), dim3(256), 0, s, in, in_ns, in_cs, out, out_ns, out_cs, C, (int)HW, N, mode, eps, weight, bias, act, res, res_ns, res_cs);
    } else if ((mode == 0 || mode == 2) && HW <= 256 * 16) {          // the plane in registers: one read instead of three (same bits as the three-pass kernel)
        hipLaunchKernelGGL(f32_norm_reg_kernel<16>, dim3(C, N), dim3(256), 0, s, in, in_ns, in_cs, out, out_ns, out_cs, C, (int)HW, mode, eps, weight, bias, act, res, res_ns, res_cs);
    } else if ((mode == 0 || mode == 2) && HW <= 256 * 64) {
        hipLaunchKernelGGL(f32_norm_reg_kernel<64>, dim3(C, N), dim3(256), 0, s, in, in_ns, in_cs, out, out_ns, out_cs, C, (int)HW, mode, eps, weight, bias, act, res, res_ns, res_cs);
    } else {
        hipLaunchKernelGGL(f32_norm_kernel, dim3(C, N), dim3(256), 0, s, in, in_ns, in_cs, out, out_ns, out_cs, C, HW, mode, eps, weight, bias, rmean, rvar, act, res, res_ns, res_cs);
    }
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

int f32_fsa_combine_launch(const float* att, int hp, int wp, int C, const float* inp, float* out, int N, int H, int W, const float* gamma, hipStream_t s, float* att_t) {
    const long tot = (long)N * C * H * W;
    if (W == 4 * wp && att_t && N <= 65535) {
        hipLaunchKernelGGL(f32_att_transpose_kernel, dim3((hp * wp + 31) / 32, (C + 31) / 32, N), dim3(256), 0, s, att, att_t, hp * wp, C);
        hipLaunchKernelGGL(f32_fsa_combine_x4_kernel, dim3((unsigned)(((long)N * C * H * (wp + 1) + 255) / 256)), dim3(256), 0, s, (const float*)att_t, hp, wp, C, inp, out, N, H, W, gamma);
    } else
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
