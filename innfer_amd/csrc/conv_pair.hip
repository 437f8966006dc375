// Two dependent 3x3 convolutions of a residual dense block in ONE tile visit (gfx950):
//     x_a = LeakyReLU(conv_a(x[0:C]))            C -> 32
//     x_b = LeakyReLU(conv_b(x[0:C] | x_a))      C + 32 -> 32
// i.e. (conv1, conv2) and (conv3, conv4) of ResidualDenseBlock_5C.forward (RRDBNet_arch.py:152-160, gc = 32, no `plus` branches).
//
// Why: layer by layer the 32-output convs are HBM-bound (224 FLOP/B, below the 315 FLOP/B ridge) and conv_b re-reads from HBM everything
// conv_a has just read plus what conv_a has just written.  Here a workgroup computes x_a on a 24 x 32 region, stores it, and goes on to x_b
// on the INNER 22 x 30 pixels of the same region while x[0:C] is still in L2 / the Infinity Cache and x_a is in the L2 it was just
// written to: HBM sees C channels read and 64 written per pixel instead of 2C + 32 read and 64 written.  The price is the one-pixel ring
// of x_a every workgroup computes for itself (768 MFMA pixels per 660 owned) and the same 24 x 32 MFMA footprint for the 22 x 30 pixels
// of x_b: 1.16 x the MFMA work of the two layers.
//
// Tile lattice: tile (i, j) OWNS pixels [22 i, 22 i + 22) x [30 j, 30 j + 30); phase A computes x_a on [22 i - 1, 22 i + 23) x [30 j - 1,
// 30 j + 31) (the halo conv_b needs; pixels outside the image are neither computed into memory nor read: the loader's zero fill is
// conv_b's zero padding), phase B computes x_b on the owned pixels.  Ring pixels of x_a are also written by the neighbouring owner --
// the same bytes (same operands, same order), so the duplicate stores are harmless, and a workgroup only ever reads back x_a it wrote itself.
//
// Machinery: conv3x3_pc<3, 2, 4> of conv3x3.hip -- 8 MFMA waves + 4 LDS-DMA waves per CU, two LDS stages of (26 x 36 px halo tile +
// 9 x 32 x 64 B weight panel), one raw barrier per 32-channel chunk, software-pipelined fragment reads -- run over 2 * C/32 + 1 chunks
// per tile: C/32 chunks of conv_a, then C/32 + 1 chunks of conv_b whose LAST chunk is x_a.  The consumers wait for their x_a stores
// (s_waitcnt vmcnt(0)) before the barrier that ends conv_b's first chunk; the loaders issue x_a's DMA at least one barrier later, with
// agent scope (sc1: served by L2, never by a stale line of the CU's vector cache).
#include "common.h"

namespace innfer {
namespace {

constexpr int TW = 32, LWP = 36, LVALID = TW + 2;
constexpr int RPW = 3, NT = 2, NLW = 4, NCW = 8;
constexpr int TH = NCW * RPW, LH = TH + 2;
constexpr int SY = TH - 2, SX = TW - 2;                   // owned pixels per tile
constexpr int NPX = LH * LWP, NQ = (NPX + 15) / 16, KQ = (NQ + NLW - 1) / NLW;
constexpr int IN_BYTES = NQ * 1024;
constexpr int WROWS = NT * 16, W_BYTES = 9 * WROWS * 64, WQ = W_BYTES / 1024;
[[maybe_unused]] constexpr int KW = (WQ + NLW - 1) / NLW;
constexpr int MT = RPW * 2;
constexpr int STAGE = IN_BYTES + W_BYTES;
constexpr int LDS_BYTES = 2 * STAGE;
constexpr int OOB = (int)0x80000000;
static_assert(LDS_BYTES <= 160 * 1024, "two stages must fit the CU's LDS");

struct PP {
    const f16* in; long in_gbytes; int nA;                // input slab (group 0), bytes between channel groups, C / 32
    const f16* wA; const float* biasA;                    // conv_a panels [chunk][tap][row][slot] (conv_pack, K = 32)
    const f16* wB; const float* biasB;                    // conv_b panels, nA + 1 chunks
    f16* outA; f16* outB;                                 // the 32-channel groups that receive x_a / x_b; outA == in + nA groups
    int N, H, W;
    int tiles_x, tiles_y, total;
    int rev;
};

__global__ __launch_bounds__(64 * (NCW + NLW), 1) void conv3x3_pair(const PP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;

    // workgroup -> tiles: every XCD owns a contiguous run of the tile list, walked round-robin by its persistent workgroups (conv3x3.hip)
    const int bid = blockIdx.x, xcd = bid & 7;
    const int run_q = p.total >> 3, run_r = p.total & 7;
    const int run_start = xcd < run_r ? xcd * (run_q + 1) : run_r * (run_q + 1) + (xcd - run_r) * run_q;
    const int run_len = run_q + (xcd < run_r ? 1 : 0);
    const int slots = ((int)gridDim.x + 7 - xcd) >> 3;
    const int per_img = p.tiles_x * p.tiles_y;
    const int j0 = bid >> 3;
    if (j0 >= run_len) return;
    const int ntiles = (run_len - j0 + slots - 1) / slots;
    const int nA = p.nA, VPT = 2 * nA + 1;                     // virtual chunks per tile: A_0 .. A_{nA-1}, B_0 .. B_{nA-1}, B_xa
    const int G = ntiles * VPT;
    const long img_elems = (long)p.H * p.W * 32;

    auto decode = [&](int jj, int& n_, int& oy_, int& ox_) {     // first OWNED pixel of the tile
        const int lid = run_start + (p.rev ? run_len - 1 - jj : jj);
        n_ = lid / per_img;
        const int tile = lid - n_ * per_img;
        const int ty = tile / p.tiles_x;
        oy_ = ty * SY;
        ox_ = (tile - ty * p.tiles_x) * SX;
    };

    if (wave >= NCW) {
        // ================================ loaders ================================
        const int lw = wave - NCW;
        int loff[KQ], voff[KQ];
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const int px = (lw + NLW * k) * 16 + (lane >> 2);
            const int ly = px / LWP, lx = px - ly * LWP;
            const int slot = (lane & 3) ^ (((px >> 2) & 1) << 1);
            loff[k] = (px < NPX && lx < LVALID) ? ((ly * p.W + lx) * 32 + slot * 8) * 2 : OOB;
        }
        const int wvoff = lane * 16;
        const char* in_tile = nullptr;
        auto setup = [&](int jj, int phase) {                  // phase 0: the tile grown by one pixel on every side
            int n, oy, ox;
            decode(jj, n, oy, ox);
            const int ty0 = oy - 1 + phase, tx0 = ox - 1 + phase;
            in_tile = (const char*)(p.in + (long)n * img_elems) + ((long)(ty0 - 1) * p.W + (tx0 - 1)) * 64;
#pragma unroll
            for (int k = 0; k < KQ; ++k) voff[k] = loff[k];
            if (ty0 <= 0 || ty0 + TH + 1 > p.H || tx0 <= 0 || tx0 + TW + 1 > p.W) {
#pragma unroll
                for (int k = 0; k < KQ; ++k) {
                    const int px = (lw + NLW * k) * 16 + (lane >> 2);
                    const int ly = px / LWP, lx = px - ly * LWP;
                    const int Y = ty0 - 1 + ly, X = tx0 - 1 + lx;
                    if (Y < 0 || Y >= p.H || X < 0 || X >= p.W) voff[k] = OOB;      // zero padding of conv_a and of conv_b
                }
            }
        };
        auto issue = [&](int v, int stage) {
#if defined(__HIP_DEVICE_COMPILE__)
            const bool phase = v >= nA;
            const int c = phase ? v - nA : v;
            const char* src = in_tile + c * p.in_gbytes;
            const char* wsrc = (const char*)(phase ? p.wB : p.wA) + (long)c * W_BYTES;
            char* st = smem + stage * STAGE;
            const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, W_BYTES, 0x00020000);
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const int jq = lw + NLW * k;
                if (jq < WQ)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(st + IN_BYTES + jq * 1024), 16, wvoff, jq * 1024, 0, 0);
            }
            if (phase && c == nA) {                           // x_a, written by this workgroup's consumers a few chunks ago: read it from L2 (sc1)
#pragma unroll
                for (int k = 0; k < KQ; ++k) {
                    const int q = lw + NLW * k;
                    if (q < NQ)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(ri, (__attribute__((address_space(3))) void*)(st + q * 1024), 16, voff[k], 0, 0, 16);
                }
            } else {
#pragma unroll
                for (int k = 0; k < KQ; ++k) {
                    const int q = lw + NLW * k;
                    if (q < NQ)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(ri, (__attribute__((address_space(3))) void*)(st + q * 1024), 16, voff[k], 0, 0, 0);
                }
            }
#else
            (void)v; (void)stage; (void)wvoff;
#endif
        };
        int jt = j0, v = 0;
        setup(jt, 0);
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        for (int g = 0; g < G; ++g) {
            if (g + 1 < G) {
                if (++v == VPT) { v = 0; jt += slots; setup(jt, 0); }
                else if (v == nA) setup(jt, 1);
                issue(v, (g + 1) & 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_barrier" ::: "memory");
        }
        return;
    }

    // ================================ consumers ================================
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
    const int aoffs = li * 64 + ((lg ^ (((li >> 2) & 1) << 1)) << 4);
    const int cbase = 4 * NT * lg;                                 // this lane's 8 output channels
    f32x4 biasA[NT], biasB[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        biasA[t] = *(const f32x4*)(p.biasA + cbase + 4 * t);
        biasB[t] = *(const f32x4*)(p.biasB + cbase + 4 * t);
    }
    f32x4 acc[NT][MT];
    int jt = j0, v = 0;
    int n = 0, oy = 0, ox = 0;
    asm volatile("s_barrier" ::: "memory");                        // chunk 0 of the first tile has landed
    for (int g = 0; g < G; ++g) {
        if (v == 0) decode(jt, n, oy, ox);
        if (v == 0 || v == nA) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[t][m] = v == 0 ? biasA[t] : biasB[t];
        }
        const char* st = smem + (g & 1) * STAGE;
        const char* sw = st + IN_BYTES;
        {
            // software-pipelined fragment reads, as in conv3x3_pc: B fragments through a three-register ring two MFMA groups ahead of their use,
            // the next tap column's weight fragments as soon as the current ones are dead
            constexpr int GPS = 2 * (RPW + 2);
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
                if (sc < 2 && seg == 1) {
                    if (rr == RPW - 1) lda(sc + 1, 0);
                    if (rr == RPW) lda(sc + 1, 1);
                    if (rr == RPW + 1) lda(sc + 1, 2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (v == nA - 1 || v == VPT - 1) {
            // LeakyReLU(0.2) -> fp16 -> one 16-byte store per pixel tile.  Phase A: every pixel of the grown tile that lies in the image;
            // phase B: the owned 22 x 30 pixels.
            const bool phB = v == VPT - 1;
            const int ty0 = oy - (phB ? 0 : 1), tx0 = ox - (phB ? 0 : 1);
            const int rlim = phB ? SY : TH, clim = phB ? SX : TW;
            f16* ob = (phB ? p.outB : p.outA) + (long)n * img_elems + cbase;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int r = cw * RPW + (m >> 1), cc = (m & 1) * 16 + li;
                const int y = ty0 + r, x = tx0 + cc;
                if (r >= rlim || cc >= clim || y < 0 || y >= p.H || x < 0 || x >= p.W) continue;
                f16x8 h;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float f = acc[t][m][j];
                        f = f > 0.f ? f : 0.2f * f;
                        h[4 * t + j] = (f16)f;
                    }
                *(f16x8*)(ob + ((long)y * p.W + x) * 32) = h;
            }
        }
        if (v == nA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // x_a is in L2 before any loader is released to fetch it
        if (++v == VPT) { v = 0; jt += slots; }
        asm volatile("s_barrier" ::: "memory");
    }
}

}  // namespace

int conv_pair_launch(const ConvPairLaunch& L, hipStream_t s) {
    if (L.C <= 0 || L.C % 32 || L.N <= 0 || L.H <= 0 || L.W <= 0) return set_error(INNFER_ERR_INVALID, "conv pair: C=%d N=%d H=%d W=%d", L.C, L.N, L.H, L.W);
    if (L.out != L.in + (long)(L.C / 32) * L.in_gstride)
        return set_error(INNFER_ERR_INVALID, "conv pair: x_a must be the channel group that follows the %d input channels (dense concat)", L.C);
    if ((long)L.N * L.H * L.W * 64 >= 0x7fffffffL) return set_error(INNFER_ERR_UNSUPPORTED, "conv pair: images of a launch must stay below 2 GiB per channel group");
    static unsigned long long attr_done = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done & bit)) {
        INNFER_HIP(hipFuncSetAttribute((const void*)conv3x3_pair, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_done |= bit;
    }
    PP p{};
    p.in = L.in; p.in_gbytes = L.in_gstride * 2; p.nA = L.C / 32;
    p.wA = L.wpk_a; p.biasA = L.bias_a; p.wB = L.wpk_b; p.biasB = L.bias_b;
    p.outA = L.out; p.outB = L.out + L.in_gstride;
    p.N = L.N; p.H = L.H; p.W = L.W;
    p.tiles_x = (L.W + SX - 1) / SX; p.tiles_y = (L.H + SY - 1) / SY;
    const long total = (long)L.N * p.tiles_x * p.tiles_y;
    if (total > 0x7fffffffL) return set_error(INNFER_ERR_INVALID, "conv pair: grid too large");
    p.total = (int)total;
    p.rev = L.rev ? 1 : 0;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const long grid = total < cus ? total : cus;
    hipLaunchKernelGGL(conv3x3_pair, dim3((unsigned)grid), dim3(64 * (NCW + NLW)), LDS_BYTES, s, p);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

}  // namespace innfer
