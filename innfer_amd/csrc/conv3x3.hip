// 3x3 stride-1 zero-pad-1 convolution as implicit GEMM on CDNA4 MFMA (gfx950).
//
// Replaces the reference's conv_block -> nn.Conv2d [+ LeakyReLU/ReLU] [+ torch.cat]
// [+ x*0.2 + residual] [+ nearest-2x Upsample in front] [+ PixelShuffle behind]
// (architectures/block.py:213-254,333-361; RRDBNet_arch.py:152-165,91-98).
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
// Workgroup = 256 threads = 4 waves, tile = (4*RPW rows) x 32 px, all 16*NT output
// channels of one channel group.  Wave w owns rows [w*RPW, (w+1)*RPW).
// Per 32-channel chunk the halo tile ((4*RPW+2) x 34 px x 64 B) and the weight
// panel (9 x 16*NT x 64 B) are staged with LDS-DMA (buffer_load_dwordx4 ... lds; out-of-image
// lanes are zero-filled by the buffer range check = the conv's zero padding);
// LDS rows are 40 px so every row base is a multiple of 8 px, which makes the
// 16-B-slot XOR swizzle (slot ^= 2*bit2(pixel)) a pure function of (lane, s):
// all ds_read_b128 are base+immediate and bank-conflict free.
// A pixel fragment B(row, seg, s) is read once and used by the up to three
// (row-in-wave, r) pairs that need it.
#include "common.h"

#include <cstdlib>

namespace innfer {

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

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
constexpr int TW = 32;        // tile width (pixels)
constexpr int LWP = 40;       // LDS row pitch (pixels), multiple of 8
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
    int N;
    unsigned tx_magic;       // ceil(2^32 / tiles_x): tile / tiles_x == (tile * tx_magic) >> 32
    int pf;                  // L2 prefetch of the next chunk's input lines
    int rev;                 // each XCD walks its run of tiles backwards
    int blk;                 // tiles are enumerated in blk x blk super-blocks (0: row-major)
    int st;                  // cache policy of the slab stores
#ifdef INNFER_ABLATE
    int abl;                 // diagnostic build only: 1 no stores, 2 no weight DMA, 4 no input DMA, 8 no MFMA phase
#endif
};

__device__ __forceinline__ void dma16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <int RPW, int NT, int OUTMODE>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma(const KP p) {
    constexpr int TH = 4 * RPW;
    constexpr int LH = TH + 2;
    constexpr int NPX = LH * LWP;
    constexpr int NQ = NPX / 16;                 // input DMA wave-instructions per chunk
    constexpr int KQ = (NQ + 3) / 4;
    constexpr int IN_BYTES = NQ * 1024;
    constexpr int WROWS = NT * 16;
    constexpr int W_BYTES = 9 * WROWS * 64;
    constexpr int WQ = W_BYTES / 1024;           // 9*NT
    constexpr int KW = (WQ + 3) / 4;
    constexpr int MT = RPW * 2;
    static_assert(NPX % 16 == 0, "tile must be a whole number of DMA pieces");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_in = smem;
    char* lds_w = smem + IN_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    STAMP(0);

    // ---- block -> (channel group, tile): XCD-aware bijective remap ------------
    // blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a
    // contiguous run of tiles so neighbouring halos and the next layer's reads
    // of the same region meet in one L2.  Speed only, never correctness.
    int lid;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int len = q + (xcd < r ? 1 : 0);
        // rev: consecutive layers walk the frame in opposite directions, so a layer starts on the
        // region its predecessor touched last -- the part of the activations that is still in the
        // 256 MiB Infinity Cache (a cyclic sweep of a larger working set would never hit).
        lid = start + (p.rev ? len - 1 - (bid >> 3) : (bid >> 3));
    }
    const int kg = lid % p.KG;
    int tile = lid / p.KG;
    const int per_img = p.tiles_x * p.tiles_y;
    const int n = tile / per_img;
    tile -= n * per_img;
    int tx, ty;
    if (p.blk > 0) {
        // super-block order: the ~64 workgroups an XCD runs at a time form a blk x blk patch of
        // tiles, so their halos meet in that XCD's L2 instead of being fetched again a tile row later
        const int B = p.blk;
        const int br = tile / (B * p.tiles_x);
        int rem = tile - br * B * p.tiles_x;
        const int hb = min(B, p.tiles_y - B * br);
        const int bc = rem / (hb * B);
        rem -= bc * hb * B;
        const int wb = min(B, p.tiles_x - B * bc);
        const int ly = rem / wb;
        ty = B * br + ly;
        tx = B * bc + (rem - ly * wb);
    } else {
        ty = tile / p.tiles_x;
        tx = tile - ty * p.tiles_x;
    }
    const int ty0 = p.y0 + ty * TH;
    const int tx0 = tx * TW;

    // ---- per-lane DMA source offsets (chunk independent) ----------------------
    const int sy_base = (ty0 > 0 ? ty0 - 1 : 0) >> p.up;
    const char* in_base = (const char*)(p.in + (long)n * p.in_img_stride +
                                        (long)sy_base * p.Ws * 32);
    int in_off[KQ];
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        const int q = wave + 4 * k;
        const int px = q * 16 + (lane >> 2);
        const int ly = px / LWP, lx = px - ly * LWP;
        const int slot = (lane & 3) ^ (((px >> 2) & 1) << 1);
        const int Y = ty0 - 1 + ly, X = tx0 - 1 + lx;
        const bool ok = (q < NQ) && (lx < LVALID) && (Y >= 0) && (Y < p.H) && (X >= 0) && (X < p.W);
        const int sy = (Y >> p.up) - sy_base, sx = X >> p.up;
        in_off[k] = ok ? ((sy * p.Ws + sx) * 32 + slot * 8) * 2 : -1;
    }
    // ---- per-lane LDS read bases ------------------------------------------------
    const int li = lane & 15, lg = lane >> 4;
    const char* bbase[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int pb = wave * RPW * LWP + li + s;
        bbase[s] = lds_in + pb * 64 + ((lg ^ ((((li + s) >> 2) & 1) << 1)) << 4);
    }
    const char* abase = lds_w + li * 64 + ((lg ^ (((li >> 2) & 1) << 1)) << 4);

    // ---- accumulators start at the bias ----------------------------------------
    const int cbase = kg * WROWS + 4 * NT * lg;     // first of this lane's 4*NT channels
    f32x4 acc[NT][MT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 b = *(const f32x4*)(p.bias + cbase + 4 * t);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[t][m] = b;
    }

    STAMP(1);
    // Issue path: every piece is ONE `buffer_load_dwordx4 ... offen lds` with a per-chunk SGPR
    // descriptor and a per-lane byte offset computed once per workgroup -- no VALU, no select.
    // Lanes whose pixel lies outside the image (the conv's zero padding) or in the LDS row pad carry
    // an offset beyond num_records: the buffer range check then writes ZEROS to LDS (verified on
    // gfx950 by the border cases of tests/test_gpu_parity.py).
    int voff[KQ];
#pragma unroll
    for (int k = 0; k < KQ; ++k) voff[k] = in_off[k] >= 0 ? in_off[k] : (int)0x80000000;
    const int wvoff = lane * 16;
    const char* w_tile = (const char*)p.wpk + (long)kg * p.nchunks * W_BYTES;

    // L2 prefetch plan (p.pf): LDS-DMA streams ~3x faster from L2 than from HBM and a workgroup can
    // only keep one chunk of LDS-DMA in flight, so while chunk c is landing / being computed every
    // thread touches one dword in (up to) two 128-B lines of chunk c+1's halo tile: ordinary loads,
    // results discarded, nothing waits for them until a whole compute phase later.
    constexpr int LPR = (LVALID * 64 + 127) / 128 + 1;        // 128-B lines per tile row (18)
    constexpr int NLINES = LH * LPR;
    int pfo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + 256 * i;
        const int r = id / LPR, l = id - r * LPR;
        const int Y = ty0 - 1 + r;
        const int xs = tx0 > 0 ? tx0 - 1 : 0;
        const bool ok = p.pf && !p.up && id < NLINES && Y >= 0 && Y < p.H && ((xs * 64) & ~127) + l * 128 < p.W * 64;
        pfo[i] = ok ? (((Y - sy_base) * p.Ws * 64 + ((xs * 64) & ~127) + l * 128)) : (int)0x80000000;
    }
    char* lds_pf = smem + IN_BYTES + W_BYTES + wave * 512;    // 2 KiB scratch nobody reads

    for (int c = 0; c < p.nchunks; ++c) {
        // stage chunk c: halo tile + weight panel, straight into LDS
#if defined(__HIP_DEVICE_COMPILE__)      // the LDS buffer-load builtin only exists in the device pass
        {
            const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(in_base + c * p.in_gbytes), 0, 0x7fffffff, 0x00020000);
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
        const bool do_pf = p.pf && !p.up && c + 1 < p.nchunks;
        if (do_pf) {
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(in_base + (c + 1) * p.in_gbytes), 0, 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (__attribute__((address_space(3))) void*)lds_pf, 4, pfo[0], 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (__attribute__((address_space(3))) void*)(lds_pf + 256), 4, pfo[1], 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // the DMA pieces, not the two prefetch loads
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#else
        (void)voff; (void)wvoff; (void)w_tile; (void)in_base; (void)KW; (void)pfo; (void)lds_pf;
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
                    const f16x8 b = *(const f16x8*)(bbase[s] + (rr * LWP + seg * 16) * 64);
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
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int y = ty0 + wave * RPW + (m >> 1);
        const int x = tx0 + (m & 1) * 16 + li;
        if (y >= p.y1 || x >= p.W) continue;
#ifdef INNFER_ABLATE
        if ((p.abl & 1) && acc[0][m][0] != 12345.678f) continue;
#endif
#ifdef INNFER_ABLATE
        // bit 16: every workgroup stores into a private 64-KiB window (cache resident, no HBM writes)
        const long pix = (p.abl & 16) ? (long)(blockIdx.x & 511) * 1024 + (wave * RPW + (m >> 1)) * 32 + (m & 1) * 16 + li
                                      : ((long)n * p.H + y) * p.W + x;
#else
        const long pix = ((long)n * p.H + y) * p.W + x;
#endif
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
                for (int j = 0; j < 4; ++j) v[t][j] = v[t][j] * p.s1 + (float)r4[j];
            }
        }
        if (p.res2) {
            const f16* rp = p.res2 + (cbase >> 5) * p.res2_gstride + pix * 32 + (cbase & 31);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f16x4 r4 = *(const f16x4*)(rp + 4 * t);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[t][j] = v[t][j] * p.s2 + (float)r4[j];
            }
        }
        if constexpr (OUTMODE == OUT_SLAB) {
            const int oc0 = cbase + p.out_coff;
                f16* op = (f16*)p.out + (oc0 >> 5) * p.out_gstride + pix * 32 + (oc0 & 31);
            if constexpr (NT >= 2) {
                // 16-byte stores; p.st picks the cache policy (speed only): 1 = sc1, 2 = sc0 sc1, 3 = nt
#pragma unroll
                for (int t = 0; t < NT; t += 2) {
                    f16x8 h;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { h[j] = (f16)v[t][j]; h[4 + j] = (f16)v[t + 1][j]; }
                    f16* o8 = op + 4 * t;
#if defined(__HIP_DEVICE_COMPILE__)
                    if (p.st == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(o8), "v"(h) : "memory");
                    else if (p.st == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(o8), "v"(h) : "memory");
                    else if (p.st == 3) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(o8), "v"(h) : "memory");
                    else
#endif
                        *(f16x8*)o8 = h;
                }
            } else {
                f16x4 h;
#pragma unroll
                for (int j = 0; j < 4; ++j) h[j] = (f16)v[0][j];
                *(f16x4*)op = h;
            }
        } else if constexpr (OUTMODE == OUT_NCHW) {
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(12);
#endif
}


struct TileCoord { int n, ty0, tx0, kg; };

int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
    }
    return n;
}

// ---------------------------------------------------------------------------------------
// v3 (INNFER_CONV_VARIANT=3): persistent single-stage kernel, 2 workgroups of 4 waves per CU.
// What the in-kernel stamps of v1 showed (profiles/r1/v1_wg_timeline.txt): a workgroup spent
// 15 % of its life in the prologue, 29 % in the epilogue, 21 % issuing LDS-DMA pieces
// (~150 cycles per 1-KiB piece, a third to a half of them weights) and only 22 % in MFMAs.
//   * persistent: grid = 2 x #CUs, every workgroup walks tiles of its XCD's contiguous range;
//     kernel-argument load, decode divisions and LDS-address setup happen once, not per tile;
//   * the next tile's first chunk is issued BEFORE the current tile's epilogue, so its DMA
//     flight (and the partner workgroup's MFMAs) cover the store tail;
//   * tile = 16 rows x 32 px for every NT: the 64-output convs re-stage their 36 KiB weight
//     panel half as often as with 8-row tiles;
//   * LDS rows are the 34 valid pixels, not padded to 40 (-13 % DMA pieces).  Row bases are
//     then only multiples of 2 px, so the swizzle bit (bit 2 of the LDS pixel index) depends
//     on e = (2*row + s) mod 8: eight per-lane base registers + immediates instead of three;
//   * bias is added in the epilogue (no global-load wait in front of the first MFMA).
// ---------------------------------------------------------------------------------------
template <int NT, int OUTMODE>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_t(const KP p) {
    constexpr int RPW = 4, NW = 4;
    constexpr int TH = NW * RPW;                 // 16
    constexpr int LH = TH + 2, LW = LVALID;      // 18 x 34
    constexpr int NPX = LH * LW;                 // 612
    constexpr int NQ = (NPX + 15) / 16;          // 39 input pieces
    constexpr int KQ = (NQ + NW - 1) / NW;       // 10
    constexpr int IN_BYTES = NQ * 1024;
    constexpr int WROWS = NT * 16;
    constexpr int W_BYTES = 9 * WROWS * 64;
    constexpr int WQ = W_BYTES / 1024;
    constexpr int KW = (WQ + NW - 1) / NW;
    constexpr int MT = RPW * 2;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_in = smem;
    char* lds_w = smem + IN_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;

    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int items = p.N * tiles_per_img * p.KG;
    int item, item_end, item_stride;
    {
        const int G = gridDim.x, b = blockIdx.x;
        const int x = b & 7, slot = b >> 3;
        const int q = items >> 3, r = items & 7;
        const int lo = x * q + (x < r ? x : r);
        const int cnt = q + (x < r ? 1 : 0);
        item_stride = (G - x + 7) >> 3;
        item = lo + slot;
        item_end = lo + cnt;
    }
    if (item >= item_end) return;

    auto decode = [&](int it) {
        TileCoord t;
        int tile = it;
        t.kg = 0;
        if (p.KG > 1) { t.kg = it % p.KG; tile = it / p.KG; }
        t.n = 0;
        if (p.N > 1) { t.n = tile / tiles_per_img; tile -= t.n * tiles_per_img; }
        const int ty = p.tiles_x == 1 ? tile
                                      : (int)(((unsigned long long)(unsigned)tile * (unsigned)p.tx_magic) >> 32);
        const int tx = tile - ty * p.tiles_x;
        t.ty0 = p.y0 + ty * TH;
        t.tx0 = tx * TW;
        return t;
    };

    // ---- per-lane constants, computed ONCE per workgroup ---------------------------
    // Tiles of one launch differ by a translation (ty0 = y0 + 16*ty, tx0 = 32*tx), so the source
    // offset of every DMA lane relative to the tile's halo origin is tile independent -- also under
    // nearest-2x upsampling, because the parity of (ty0-1, tx0-1) is the same for all tiles.
    const int par_y = p.up ? ((p.y0 - 1) & 1) : 0;
    const int par_x = p.up ? 1 : 0;                        // tx0 - 1 is odd
    int loff[KQ];                                          // byte offset from the halo origin, -1 = never valid
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        const int q = wave + NW * k;
        const int px = q * 16 + (lane >> 2);
        const int ly = px / LW, lx = px - ly * LW;
        const int slot = (lane & 3) ^ (((px >> 2) & 1) << 1);
        const int ry = p.up ? (par_y + ly) >> 1 : ly;
        const int rx = p.up ? (par_x + lx) >> 1 : lx;
        loff[k] = (q < NQ && px < NPX) ? ((ry * p.Ws + rx) * 32 + slot * 8) * 2 : -1;
    }
    const int woff = lane * 16;

    int bo[8];                                             // LDS read offsets, one per e = (2*rr + s) & 7
#pragma unroll
    for (int e = 0; e < 8; ++e)
        bo[e] = (wave * RPW * LW + li) * 64 + ((lg ^ ((((e + li) >> 2) & 1) << 1)) << 4);
    const int ao = IN_BYTES + li * 64 + ((lg ^ (((li >> 2) & 1) << 1)) << 4);

    // ---- per-tile state ---------------------------------------------------------------
    const char* t_base = nullptr;                          // address of the halo origin, chunk 0
    const char* w_base = nullptr;
    unsigned vmask = 0;                                    // bit k: this lane's piece k reads real data
    auto setup = [&](const TileCoord& t) {
        const int oy = (t.ty0 - 1) >> p.up, ox = (t.tx0 - 1) >> p.up;       // arithmetic shift = floor
        t_base = (const char*)p.in + ((long)t.n * p.in_img_stride + ((long)oy * p.Ws + ox) * 32) * 2;
        w_base = (const char*)p.wpk + (long)t.kg * p.nchunks * W_BYTES + woff;
        const bool interior = t.ty0 >= 1 && t.ty0 + TH < p.H && t.tx0 >= 1 && t.tx0 + TW < p.W;
        vmask = 0;
        if (interior) {
#pragma unroll
            for (int k = 0; k < KQ; ++k) vmask |= (loff[k] >= 0 ? 1u : 0u) << k;
        } else {
#pragma unroll
            for (int k = 0; k < KQ; ++k) {
                const int px = (wave + NW * k) * 16 + (lane >> 2);
                const int ly = px / LW, lx = px - ly * LW;
                const int Y = t.ty0 - 1 + ly, X = t.tx0 - 1 + lx;
                const bool ok = loff[k] >= 0 && Y >= 0 && Y < p.H && X >= 0 && X < p.W;
                vmask |= (ok ? 1u : 0u) << k;
            }
        }
    };
    auto issue = [&](int c) {
        const char* cb = t_base + c * p.in_gbytes;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const int q = wave + NW * k;
            if (q < NQ) {
                const char* src = (vmask >> k) & 1 ? cb + loff[k] : (const char*)g_zero_page;
                dma16(src, lds_in + q * 1024);
            }
        }
        const char* wb = w_base + (long)c * W_BYTES;
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const int j = wave + NW * k;
            if (j < WQ) dma16(wb + j * 1024, lds_w + j * 1024);
        }
    };

    TileCoord cur = decode(item);
    setup(cur);
    issue(0);

    // NT <= 2: the bias of this lane's channels lives in registers (accumulators start from it) and
    // is reloaded only when the channel group changes; NT == 4 is register-bound (128 accumulators +
    // 48 weight-fragment registers) and adds the bias in the epilogue instead.
    constexpr bool BIAS_REGS = NT <= 2;
    constexpr int NB = BIAS_REGS ? NT : 1;
    int bias_kg = cur.kg;
    f32x4 bias[NB];
    if constexpr (BIAS_REGS) {
#pragma unroll
        for (int t = 0; t < NT; ++t) bias[t] = *(const f32x4*)(p.bias + cur.kg * WROWS + 4 * NT * lg + 4 * t);
    }

    while (true) {
        if constexpr (BIAS_REGS) {
            if (p.KG > 1 && cur.kg != bias_kg) {
                bias_kg = cur.kg;
#pragma unroll
                for (int t = 0; t < NT; ++t) bias[t] = *(const f32x4*)(p.bias + cur.kg * WROWS + 4 * NT * lg + 4 * t);
            }
        }
        f32x4 acc[NT][MT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if constexpr (BIAS_REGS) acc[t][m] = bias[t];
                else acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
            }

        const int next_item = item + item_stride;
        const bool has_next = next_item < item_end;
        TileCoord nxt = cur;

        for (int c = 0; c < p.nchunks; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                f16x8 a[3][NT];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        a[r][t] = *(const f16x8*)(smem + ao + ((r * 3 + s) * WROWS + t * 16) * 64);
#pragma unroll
                for (int rr = 0; rr < RPW + 2; ++rr) {
#pragma unroll
                    for (int seg = 0; seg < 2; ++seg) {
                        const f16x8 b = *(const f16x8*)(smem + bo[(2 * rr + s) & 7] + (rr * LW + seg * 16 + s) * 64);
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
            __syncthreads();
            // LDS is free again: stage the next chunk -- or the NEXT TILE's first chunk, whose
            // flight then overlaps this tile's epilogue.
            if (c + 1 < p.nchunks) {
                issue(c + 1);
            } else if (has_next) {
                nxt = decode(next_item);
                setup(nxt);
                issue(0);
            }
        }

        // ---- epilogue of tile `cur`: act -> residuals -> store ------------------------
        {
            const int cbase = cur.kg * WROWS + 4 * NT * lg;
            const int y_w = cur.ty0 + wave * RPW;                      // first row of this wave
            const long pix0 = ((long)cur.n * p.H + y_w) * p.W + cur.tx0 + li;
            const bool full = (y_w + RPW <= p.y1) && (cur.tx0 + TW <= p.W);
            f32x4 lbias[NT];
            if constexpr (!BIAS_REGS) {
#pragma unroll
                for (int t = 0; t < NT; ++t) lbias[t] = *(const f32x4*)(p.bias + cbase + 4 * t);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (!full) {
                    if (y_w + (m >> 1) >= p.y1 || cur.tx0 + (m & 1) * 16 + li >= p.W) continue;
                }
                const long pix = pix0 + (long)(m >> 1) * p.W + (m & 1) * 16;
                float v[NT][4];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float f = acc[t][m][j];
                        if constexpr (!BIAS_REGS) f += lbias[t][j];
                        v[t][j] = p.act == 1 ? fmaxf(f, 0.2f * f) : (p.act == 2 ? fmaxf(f, 0.f) : f);
                    }
                if (p.res1) {
                    const f16* rp = p.res1 + (cbase >> 5) * p.res1_gstride + pix * 32 + (cbase & 31);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const f16x4 r4 = *(const f16x4*)(rp + 4 * t);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[t][j] = v[t][j] * p.s1 + (float)r4[j];
                    }
                }
                if (p.res2) {
                    const f16* rp = p.res2 + (cbase >> 5) * p.res2_gstride + pix * 32 + (cbase & 31);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const f16x4 r4 = *(const f16x4*)(rp + 4 * t);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[t][j] = v[t][j] * p.s2 + (float)r4[j];
                    }
                }
                if constexpr (OUTMODE == OUT_SLAB) {
                    const int oc0 = cbase + p.out_coff;
                    f16* op = (f16*)p.out + (oc0 >> 5) * p.out_gstride + pix * 32 + (oc0 & 31);
                    f16 h[4 * NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int j = 0; j < 4; ++j) h[4 * t + j] = (f16)v[t][j];
                    if constexpr (NT == 1) {
                        *(f16x4*)op = *(const f16x4*)h;
                    } else {
#pragma unroll
                        for (int u = 0; u < NT / 2; ++u) *(f16x8*)(op + 8 * u) = *(const f16x8*)(h + 8 * u);
                    }
                } else if constexpr (OUTMODE == OUT_NCHW) {
                    const int y = y_w + (m >> 1), x = cur.tx0 + (m & 1) * 16 + li;
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int ch = cbase + 4 * t + j;
                            if (ch < p.K) {
                                const long o = (((long)cur.n * p.K + ch) * p.H + y) * p.W + x;
                                if (p.out_f32) ((float*)p.out)[o] = v[t][j];
                                else ((f16*)p.out)[o] = (f16)v[t][j];
                            }
                        }
                } else {
                    const int y = y_w + (m >> 1), x = cur.tx0 + (m & 1) * 16 + li;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const long opix = ((long)cur.n * 2 * p.H + 2 * y + (j >> 1)) * (2 * p.W) + 2 * x + (j & 1);
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
        }
        if (!has_next) break;
        cur = nxt;
        item = next_item;
    }
}

template <int NT, int OUTMODE>
int launch_v3(const KP& kp, int N, hipStream_t s) {
    constexpr int LDS = ((18 * LVALID + 15) / 16) * 1024 + 9 * NT * 16 * 64;
    static bool attr_done = false;
    if (!attr_done) {
        INNFER_HIP(hipFuncSetAttribute((const void*)conv3x3_mfma_t<NT, OUTMODE>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_done = true;
    }
    KP k = kp;
    k.N = N;
    k.tiles_x = (k.W + TW - 1) / TW;
    k.tiles_y = (k.y1 - k.y0 + 15) / 16;
    k.tx_magic = k.tiles_x > 1 ? (unsigned)(0x100000000ULL / (unsigned)k.tiles_x + 1) : 0;   // exact for tile < 2^24
    const long items = (long)N * k.tiles_x * k.tiles_y * k.KG;
    if (items <= 0) return INNFER_OK;
    if (items > 0x7fffffffL || (long)k.tiles_x * k.tiles_y >= (1L << 24))
        return set_error(INNFER_ERR_INVALID, "conv grid too large");
    const int slots = 2 * num_cus();
    const int grid = (int)(items < slots ? items : slots);
    hipLaunchKernelGGL((conv3x3_mfma_t<NT, OUTMODE>), dim3(grid), dim3(256), LDS, s, k);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

int conv_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("INNFER_CONV_VARIANT");
        v = e ? atoi(e) : 1;
    }
    return v;
}

template <int RPW, int NT, int OUTMODE>
int launch_t(const KP& kp, int N, hipStream_t s) {
    constexpr int TH = 4 * RPW;
    constexpr int LDS = ((TH + 2) * LWP / 16) * 1024 + 9 * NT * 16 * 64 + 2048;   // + prefetch scratch
    static bool attr_done = false;
    if (!attr_done) {
        INNFER_HIP(hipFuncSetAttribute((const void*)conv3x3_mfma<RPW, NT, OUTMODE>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_done = true;
    }
    KP k = kp;
    k.tiles_x = (k.W + TW - 1) / TW;
    k.tiles_y = (k.y1 - k.y0 + TH - 1) / TH;
    {
        static const int pf = getenv("INNFER_PREFETCH") ? atoi(getenv("INNFER_PREFETCH")) : 1;
        static const int blk = getenv("INNFER_TILE_BLOCK") ? atoi(getenv("INNFER_TILE_BLOCK")) : 0;
        static const int st = getenv("INNFER_STORE") ? atoi(getenv("INNFER_STORE")) : 0;
        k.pf = pf;
        k.blk = blk;
        k.st = st;
#ifdef INNFER_ABLATE
        k.abl = getenv("INNFER_ABL") ? atoi(getenv("INNFER_ABL")) : 0;
#endif
    }
    const long grid = (long)N * k.tiles_x * k.tiles_y * k.KG;
    if (grid <= 0) return INNFER_OK;
    if (grid > 0x7fffffffL) return set_error(INNFER_ERR_INVALID, "conv grid too large");
    hipLaunchKernelGGL((conv3x3_mfma<RPW, NT, OUTMODE>), dim3((unsigned)grid), dim3(256), LDS, s, k);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

}  // namespace

}  // namespace innfer
extern "C" int innfer_debug_read_stamps(unsigned long long* h, int n_words) {
#ifdef INNFER_STAMPS
    return hipMemcpyFromSymbol(h, HIP_SYMBOL(innfer::g_stamps), (size_t)n_words * 8) == hipSuccess ? NSTAMP : -2;
#else
    (void)h; (void)n_words;
    return 0;
#endif
}
namespace innfer {

int conv_nt_for(int K) { return K >= 64 ? 4 : (K >= 32 ? 2 : 1); }

static int conv_groups(int K) {
    const int per = 16 * conv_nt_for(K);
    return (K + per - 1) / per;
}

size_t conv_packed_bytes(int K, int C) {
    const int nt = conv_nt_for(K);
    return (size_t)conv_groups(K) * (C / 32) * 9 * nt * 16 * 64;
}

// Host: OIHW fp32 -> [group][chunk][tap][row R][slot][8 ch] fp16, the exact LDS image.
// Row R = t*16 + rho of a group holds out channel  group*16*NT + (4*NT)*(rho>>2) + 4*t + (rho&3);
// slot sigma holds input channels chunk*32 + 8*(sigma ^ 2*bit2(R)) .. +7.
void conv_pack(const float* w, int K, int C, void* packed) {
    const int nt = conv_nt_for(K), rows = nt * 16, groups = conv_groups(K), nch = C / 32;
    f16* dst = (f16*)packed;
    for (int g = 0; g < groups; ++g)
        for (int c = 0; c < nch; ++c)
            for (int tap = 0; tap < 9; ++tap)
                for (int R = 0; R < rows; ++R) {
                    const int t = R >> 4, rho = R & 15;
                    const int oc = g * rows + 4 * nt * (rho >> 2) + 4 * t + (rho & 3);
                    for (int sg = 0; sg < 4; ++sg) {
                        const int cg = sg ^ (((R >> 2) & 1) << 1);
                        for (int e = 0; e < 8; ++e) {
                            const int ic = c * 32 + cg * 8 + e;
                            const float v = oc < K ? w[((size_t)oc * C + ic) * 9 + tap] : 0.f;
                            *dst++ = (f16)v;
                        }
                    }
                }
}

int conv_launch(const ConvLaunch& L, hipStream_t s) {
    if (L.C <= 0 || L.C % 32) return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: C=%d must be a multiple of 32", L.C);
    if (L.up && ((L.H | L.W) & 1)) return set_error(INNFER_ERR_INVALID, "conv3x3: upsampled size must be even");
    KP k{};
    k.up = L.up ? 1 : 0;
    k.H = L.H; k.W = L.W; k.Hs = L.H >> k.up; k.Ws = L.W >> k.up;
    k.in = L.in; k.in_gbytes = L.in_gstride * 2; k.in_img_stride = (long)k.Hs * k.Ws * 32;
    k.nchunks = L.C / 32;
    k.wpk = L.wpk; k.bias = L.bias;
    k.out = L.out; k.out_gstride = L.out_gstride; k.out_coff = L.out_coff;
    k.K = L.K; k.KG = conv_groups(L.K);
    k.act = L.act;
    k.res1 = L.res1; k.res1_gstride = L.res1_gstride; k.s1 = L.s1;
    k.res2 = L.res2; k.res2_gstride = L.res2_gstride; k.s2 = L.s2;
    k.y0 = L.y0; k.y1 = L.y1 > 0 ? L.y1 : L.H;
    if (k.y0 < 0 || k.y1 > L.H || k.y0 >= k.y1) return set_error(INNFER_ERR_INVALID, "conv3x3: bad row range [%d,%d)", k.y0, k.y1);
    k.out_f32 = L.out_f32;
    k.rev = L.rev ? 1 : 0;
    const int nt = conv_nt_for(L.K);
    if (L.out_mode != OUT_NCHW && L.K % (16 * nt))
        return set_error(INNFER_ERR_UNSUPPORTED, "conv3x3: K=%d must be a multiple of %d for slab output", L.K, 16 * nt);
    if (conv_variant() == 3) {
        switch (L.out_mode) {
            case OUT_SLAB:
                if (nt == 4) return launch_v3<4, OUT_SLAB>(k, L.N, s);
                if (nt == 2) return launch_v3<2, OUT_SLAB>(k, L.N, s);
                return launch_v3<1, OUT_SLAB>(k, L.N, s);
            case OUT_NCHW:
                if (nt == 4) return launch_v3<4, OUT_NCHW>(k, L.N, s);
                if (nt == 2) return launch_v3<2, OUT_NCHW>(k, L.N, s);
                return launch_v3<1, OUT_NCHW>(k, L.N, s);
            case OUT_SHUFFLE2:
                if (nt == 4) return launch_v3<4, OUT_SHUFFLE2>(k, L.N, s);
                return set_error(INNFER_ERR_UNSUPPORTED, "pixelshuffle conv needs K %% 64 == 0");
        }
        return set_error(INNFER_ERR_INVALID, "conv3x3: bad out_mode");
    }
    static const int rpw64 = getenv("INNFER_RPW64") ? atoi(getenv("INNFER_RPW64")) : 3;
    static const int rpw32 = getenv("INNFER_RPW32") ? atoi(getenv("INNFER_RPW32")) : 5;
    switch (L.out_mode) {
        case OUT_SLAB:
            if (nt == 4) return rpw64 == 3 ? launch_t<3, 4, OUT_SLAB>(k, L.N, s) : launch_t<2, 4, OUT_SLAB>(k, L.N, s);
            if (nt == 2) return rpw32 == 5 ? launch_t<5, 2, OUT_SLAB>(k, L.N, s) : launch_t<4, 2, OUT_SLAB>(k, L.N, s);
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
