// One SCPA block of PAN (reference architectures/PAN_arch.py:58-105) in the fp32-accurate mode (`-no_fp16`, run.py:345,421-422) as ONE launch on the fp16 matrix
// cores: the split-operand form of csrc/pan_scpa.hip (round 6).
//
// Every tensor is a PAIR of fp16 values per element, hi = fp16(x) and lo = fp16((x - hi) * 2^11) (22 significant bits; the convention of the SR engine's SPLIT mode,
// conv3x3_epilogue_slab.h), every weight likewise, and a product is xh * wh + 2^-11 (xh * wl + xl * wh) (the 2^-22 term is dropped): three MFMAs per k-step into two
// fp32 accumulators (main, cross), joined as main + 2^-11 cross before the activation.  The block's intermediates A | B, a', Y, b' never leave the CU; they are re-split
// where the fp16 kernel rounds them to fp16.
//
// Differences to the fp16 kernel, all forced by LDS (every pixel costs twice the bytes, the weights too: 2 x 34 KB):
//   * 8 x 32 pixel tiles (12 x 36 halo pixels), four LDS planes A_hi | A_lo | B_hi | B_lo of 48 B per pixel (83 KB);
//   * x is NOT staged in LDS: conv1_a / conv1_b are 1x1, so a wave reads the B operand of its pixel tiles straight from global memory in fragment layout (a lane: 8
//     consecutive channels of a pixel = one 16-byte load), the NEXT tile's during this tile's P3; the residual x of the wave's own pixels is re-read (L2) in P3;
//   * a wave of P2a / P3 owns two rows of one 16-pixel segment (a column's four input rows feed both: 20 fragment reads per 36 MFMAs instead of 24).
// LDS layouts chosen for the bank rules of ds_read_b128 (four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, .. -- each needing 16 distinct 16-byte slots
// mod 256 B; the fp16 kernel's pixel-major 48-byte pixels and row-major panels put two lanes of a group on one slot: 0.41 of the LDS cycles of the first version were
// conflict cycles, 18 % of the launch, profiles/r6/pmc_panf32.txt):
//   * pixel planes OCTET-major: octet o of pixel P at o * 6912 + P * 16 (6912 = 27 x 256): a group's 16 lanes read 16 consecutive pixels of one or two octet planes;
//     lanes of k-octet 3 (structural padding) read octet 2 of their own pixel: finite data in a slot of their own, against weights that are zero there;
//   * the 20 -> 20 convs' panels per tap as [12 rows of tile 0 | 8 rows of tile 1 | zero | zero] with the rows of a tile in a PERMUTED order (3 sigma(row) + octet mod 16
//     distinct within every group); lanes of rows beyond the real ones read the fragment of a real row of their own group (a broadcast; their output rows are channels
//     nothing reads with a non-zero weight), lanes of k-octet 3 the tap's zero slot -- whose residue no real fragment of their group has.  Every fragment address is
//     lane constant + immediate: no select, no address arithmetic per read.
// Tensor layout in memory ("split planes", npx = N * H * W pixels): [hi, channels 0..31: 64 B per pixel][hi, 32..39: 16 B][lo, 0..31: 64 B][lo, 32..39: 16 B].
#include <atomic>
#include <type_traits>
#include <vector>
#include "common.h"
#include "pan_scpa_layout.h"

namespace innfer {

namespace {

using namespace scpa;

constexpr int TH = 8, HR = TH + 4, NPX = HR * HC, NP1 = NPX / 16;          // 12 x 36 halo pixels = 27 pixel tiles of 16
constexpr int YW = TW + 2, NY = (TH + 2) * YW, NMID = (NY + 15) / 16;        // Y: the 10 x 34 region k4 reads = 22 pixel tiles
constexpr int KT = 4;                                                       // P1 tile slots per wave (p1_tile)
static_assert(NP1 == 27, "p1_tile's work list");
constexpr int WLO = (W_BYTES + 255) / 256 * 256;                            // the lo blob behind the hi blob
constexpr int PL = NPX * 16;                                                // an octet plane of the LDS images
constexpr int AH = 2 * WLO, AL = AH + 3 * PL, BH = AL + 3 * PL, BL = BH + 3 * PL, TAIL = BL + 3 * PL;
constexpr int LDS_BYTES = TAIL + 1024;
static_assert(NPX % 16 == 0 && PL % 256 == 0 && LDS_BYTES <= 160 * 1024, "LDS");
static_assert(WLO + OFF_K2 + K_TAP < 65536 && 3 * PL + (3 * HC + 2) * 16 < 65536, "fragment offsets fit the 16-bit immediate of ds_read_b128");
constexpr int OOB = (int)0x80000000;
constexpr float UP = 2048.0f, DOWN = 1.0f / 2048.0f;

struct SplitKP {
    const char* in; char* out;              // split planes (see above)
    const char* w;                          // hi blob | lo blob (2 * WLO bytes)
    int npx;                                // N * H * W
    int N, H, W, tiles_x, tiles_y, total;
};

__device__ __forceinline__ f16x8 lds16(const char* smem, int off, bool real) { return *(const f16x8*)(smem + (real ? off : OFF_ZERO)); }
// P1's work list: 27 pixel tiles of the halo region, each an A unit and a B unit.  Slot k < 3 of wave w: tile w + 8 k (both units); slot 3: tiles 24 .. 26, the A unit by
// waves 0 .. 2 and the B unit by waves 3 .. 5 (3.5 tile slots on the longest wave instead of 4: the phase ends at the barrier behind its slowest wave)
__device__ __forceinline__ int p1_tile(int wave, int k) { return k < 3 ? wave + 8 * k : (wave < 6 ? 24 + (wave < 3 ? wave : wave - 3) : -1); }
__device__ __forceinline__ f16x8 px16(const char* smem, int off) { return *(const f16x8*)(smem + off); }
__device__ __forceinline__ f16x8 gload(const __amdgpu_buffer_rsrc_t rs, int off) { return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0)); }

// v = main + 2^-11 cross (+ LeakyReLU) of a lane's eight channels -> (hi, lo)
template <bool ACT>
__device__ __forceinline__ void join_split8(const f32x4 (&m)[2], const f32x4 (&x)[2], f16x8& h, f16x8& l) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float f = __builtin_fmaf(x[t][j], DOWN, m[t][j]);
            if (ACT) f = __builtin_amdgcn_fmed3f(f, 0.2f * f, 3.0e38f);       // (LeakyReLU; a finite top: with +inf LLVM emits max(x, 0.2 x) behind a canonicalising v_max x, x)
            const f16 hh = (f16)f;
            h[4 * t + j] = hh;
            l[4 * t + j] = (f16)__builtin_fmaf((float)hh, -UP, f * UP);          // = (f - hh) * 2^11 exactly: one v_fma_mixlo_f16 behind the multiply
        }
}
__device__ __forceinline__ void split8(const float (&f)[8], f16x8& h, f16x8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const f16 hh = (f16)f[j];
        h[j] = hh;
        l[j] = (f16)__builtin_fmaf((float)hh, -UP, f[j] * UP);
    }
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)

// A 20 -> 20 channel 3x3 conv over the wave's two rows of one segment of the LDS image `src` (hi planes; the lo planes 3 PL behind): pb = the lane's pixel-fragment
// base (plane of its octet + its pixel of the tile's first tap), w0 / w1 = the lane's fragment offsets inside a tap block (tiles 0 / 1), woff = the conv's panels in the hi
// blob (WLO further: the lo blob).  Column by column; a column's twelve weight fragments and its four input rows' hi / lo pixel fragments are read up front.
__device__ __forceinline__ void conv33_split(const char* smem, int pb, int w0, int w1, int woff, f32x4 (&cm)[2][2], f32x4 (&cx)[2][2]) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        f16x8 wh[3][2], wl[3][2], bh[4], bl[4];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int o = (t ? w1 : w0) + woff + (dy * 3 + dx) * K_TAP;
                wh[dy][t] = px16(smem, o);
                wl[dy][t] = px16(smem, o + WLO);
            }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            bh[rr] = px16(smem, pb + (rr * HC + dx) * 16);
            bl[rr] = px16(smem, pb + (rr * HC + dx) * 16 + 3 * PL);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int lo = rr - dy;
                if (lo >= 0 && lo < 2) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        cm[lo][t] = MFMA(wh[dy][t], bh[rr], cm[lo][t]);
                        cx[lo][t] = MFMA(wl[dy][t], bh[rr], cx[lo][t]);
                        cx[lo][t] = MFMA(wh[dy][t], bl[rr], cx[lo][t]);
                    }
                }
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ __launch_bounds__(512, 1) void pan_scpa_split(const SplitKP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li_w = lane & 15, lg_w = lane >> 4;

    // the XCD's workgroups (blocks b, b + 8, ..) walk that XCD's contiguous run of the tile list: halos meet in one L2
    const int bid = blockIdx.x, xcd = bid & 7;
    const int run_q = p.total >> 3, run_r = p.total & 7;
    const int run_start = xcd < run_r ? xcd * (run_q + 1) : run_r * (run_q + 1) + (xcd - run_r) * run_q;
    const int run_len = run_q + (xcd < run_r ? 1 : 0);
    const int slots = ((int)gridDim.x + 7 - xcd) >> 3;
    const int j0 = bid >> 3;
    if (j0 >= run_len) return;
    const int per_img = p.tiles_x * p.tiles_y;
    const int o_hi8 = p.npx * 64, o_lo32 = p.npx * 80, o_lo8 = p.npx * 144;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.npx * 160, 0x00020000);

    auto decode = [&](int j, int& n, int& ty0, int& tx0) __attribute__((always_inline)) {
        int t = run_start + j;
        n = t / per_img; t -= n * per_img;
        const int ty = t / p.tiles_x;
        ty0 = ty * TH; tx0 = (t - ty * p.tiles_x) * TW;
    };
    // the B operand of this wave's P1 pixel tiles (halo pixels 16 (wave + 8 k) + li of tile j), straight from memory: [0] hi 0..31 (octet lg), [1] hi 32..39,
    // [2] lo 0..31, [3] lo 32..39; pixels outside the image read zero (the buffer's range check) = the zero padding of every conv of the block
    f16x8 xh0[KT], xh1[KT], xl0[KT], xl1[KT];
    auto load_x = [&](int j) __attribute__((always_inline)) {
        int n, ty0, tx0;
        decode(j, n, ty0, tx0);
        const int base = (n * p.H + ty0 - 2) * p.W + tx0 - 2;
        const bool edge = ty0 < 2 || ty0 + TH + 2 > p.H || tx0 < 2 || tx0 + TW + 2 > p.W;
        int li = li_w, lg = lg_w;
        asm volatile("" : "+v"(li), "+v"(lg));
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const int P = 16 * p1_tile(wave, k) + li, r = P / HC, c = P - r * HC;
            int o32 = (base + r * p.W + c) * 64 + lg * 16, o8 = (base + r * p.W + c) * 16;
            if (edge) {
                const int y = ty0 - 2 + r, x = tx0 - 2 + c;
                if (y < 0 || y >= p.H || x < 0 || x >= p.W) o32 = o8 = OOB;
            }
            if (p1_tile(wave, k) < 0) o32 = o8 = OOB;         // (an empty slot: the range check answers without a memory access)
            xh0[k] = gload(rs, o32);
            xh1[k] = gload(rs, o8 + o_hi8);
            xl0[k] = gload(rs, o32 + o_lo32);
            xl1[k] = gload(rs, o8 + o_lo8);
        }
    };

    load_x(j0);
#if defined(__HIP_DEVICE_COMPILE__)
    {   // the two blobs global -> LDS by LDS-DMA: every wave issues its 1-KB pieces back to back (one round trip instead of a load -> store loop's several)
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 2 * WLO, 0x00020000);
        for (int q = wave; q * 1024 < 2 * WLO; q += 8)
            if (q * 1024 + lane * 16 < 2 * WLO)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(smem + q * 1024), 16, q * 1024 + lane * 16, 0, 0, 0);
    }
#endif
    // (finite data wherever a fragment read may land: P1 writes every pixel of all twelve planes before the first read, so only the 1 KB behind them -- where the reads of
    //  the last pixels' right-hand taps end -- needs initialising)
    if (tid < 64) *(f16x8*)(smem + TAIL + tid * 16) = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const int row0 = 2 * (wave >> 1), seg = wave & 1;        // this wave's two output rows / its segment (P2a, P3)
    // lane constants of the fragment reads (see the head of the file)
    const int lo_w = lg_w < 3 ? lg_w : 2;                                                    // the octet plane a lane reads (k-octet 3: octet 2 again)
    const int w0_w = kfrag_t0(li_w, lg_w), w1_w = kfrag_t1(li_w, lg_w);
    const int pconv_w = lo_w * PL + ((row0 + 1) * HC + 1 + seg * 16 + li_w) * 16;              // + AH / AL ..: the first tap's pixel of the wave's first row
    for (int j = j0; j < run_len; j += slots) {
        int n, ty0, tx0;
        decode(j, n, ty0, tx0);
        int li = li_w, lg = lg_w;
        asm volatile("" : "+v"(li), "+v"(lg));
        // ---------------- P1: A | B = lrelu(conv1_a | conv1_b (x)) over the halo tile ----------------
        {
            f16x8 wh[2][2][2], wl[2][2][2];                  // [a | b][t][k-step]
#pragma unroll
            for (int ab = 0; ab < 2; ++ab)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int b = (ab ? OFF_C1B : OFF_C1A) + (t ? C1_T1 : 0) + li * C1_ROW;
                    wh[ab][t][0] = lds16(smem, b + lg * 16, li < r2(t)); wh[ab][t][1] = lds16(smem, b + 64, li < r2(t) && lg == 0);
                    wl[ab][t][0] = lds16(smem, WLO + b + lg * 16, li < r2(t)); wl[ab][t][1] = lds16(smem, WLO + b + 64, li < r2(t) && lg == 0);
                }
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                if (p1_tile(wave, k) < 0) continue;
                const int P = 16 * p1_tile(wave, k) + li;
#pragma unroll
                for (int ab = 0; ab < 2; ++ab) {
                    if (k == KT - 1 && ab != (wave >= 3 ? 1 : 0)) continue;      // (the last three pixel tiles: A by waves 0..2, B by waves 3..5)
                    f32x4 cm[2], cx[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        cm[t] = MFMA(wh[ab][t][0], xh0[k], z4);
                        cm[t] = MFMA(wh[ab][t][1], xh1[k], cm[t]);
                        cx[t] = MFMA(wl[ab][t][0], xh0[k], z4);
                        cx[t] = MFMA(wl[ab][t][1], xh1[k], cx[t]);
                        cx[t] = MFMA(wh[ab][t][0], xl0[k], cx[t]);
                        cx[t] = MFMA(wh[ab][t][1], xl1[k], cx[t]);
                    }
                    f16x8 h, l;
                    join_split8<true>(cm, cx, h, l);
                    if (lg < 3) {
                        *(f16x8*)(smem + (ab ? BH : AH) + lg * PL + P * 16) = h;
                        *(f16x8*)(smem + (ab ? BL : AL) + lg * PL + P * 16) = l;
                    }
                }
            }
        }
        __syncthreads();
        // ---------------- P2a: a' = lrelu(k1(A)) on the wave's own pixels (registers: the B operand of conv3) ----------------
        f16x8 aph[2], apl[2];
        {
            f32x4 cm[2][2], cx[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) cm[u][0] = cm[u][1] = cx[u][0] = cx[u][1] = z4;
            conv33_split(smem, AH + pconv_w, w0_w, w1_w, OFF_K1, cm, cx);
#pragma unroll
            for (int u = 0; u < 2; ++u) join_split8<true>(cm[u], cx[u], aph[u], apl[u]);
        }
        __syncthreads();                                            // every wave has read A: Y may take its place
        // ---------------- P2b: Y = k3(B) * sigmoid(k2(B) + bias) on rows 1 .. TH + 2, zero outside the image ----------------
        {
            const f32x4 bk0 = *(const f32x4*)(smem + OFF_B2 + (8 * lg) * 4), bk1 = *(const f32x4*)(smem + OFF_B2 + (8 * lg + 4) * 4);
            const bool edge_t = ty0 < 2 || ty0 + TH + 2 > p.H || tx0 < 2 || tx0 + TW + 2 > p.W;
            auto halo_px = [&](int i) __attribute__((always_inline)) {
                const int Q = min(16 * i + li, NY - 1), r = Q / YW;
                return (r + 1) * HC + (Q - r * YW) + 1;
            };
            auto pass = [&](auto nkc, int i0) __attribute__((always_inline)) {      // NK pixel tiles i0, i0 + 8, .. of this wave, their fragment reads issued together
                constexpr int NK = decltype(nkc)::value;
                f32x4 cm[NK][2], cx[NK][2], gm[NK][2], gx[NK][2];
                int Pk[NK];
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    cm[k][0] = cm[k][1] = cx[k][0] = cx[k][1] = gx[k][0] = gx[k][1] = z4;
                    gm[k][0] = bk0; gm[k][1] = bk1;
                    Pk[k] = halo_px(i0 + 8 * k < NMID ? i0 + 8 * k : i0);
                }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    f16x8 wh[3][2], wl[3][2], bh[NK][3], bl[NK][3];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const int o = (t ? w1_w : w0_w) + OFF_K3 + (dy * 3 + dx) * K_TAP;
                            wh[dy][t] = px16(smem, o);
                            wl[dy][t] = px16(smem, o + WLO);
                        }
#pragma unroll
                    for (int k = 0; k < NK; ++k)
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int o = BH + lo_w * PL + (Pk[k] - HC - 1) * 16 + (dy * HC + dx) * 16;
                            bh[k][dy] = px16(smem, o);
                            bl[k][dy] = px16(smem, o + 3 * PL);
                        }
#pragma unroll
                    for (int k = 0; k < NK; ++k)
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int t = 0; t < 2; ++t) {
                                cm[k][t] = MFMA(wh[dy][t], bh[k][dy], cm[k][t]);
                                cx[k][t] = MFMA(wl[dy][t], bh[k][dy], cx[k][t]);
                                cx[k][t] = MFMA(wh[dy][t], bl[k][dy], cx[k][t]);
                            }
                    if (dx == 1) {                                  // k2: the 1x1 conv of the gate reads the centre pixel's fragments
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const int o = (t ? w1_w : w0_w) + OFF_K2;
                            const f16x8 w2h = px16(smem, o), w2l = px16(smem, o + WLO);
#pragma unroll
                            for (int k = 0; k < NK; ++k) {
                                gm[k][t] = MFMA(w2h, bh[k][1], gm[k][t]);
                                gx[k][t] = MFMA(w2l, bh[k][1], gx[k][t]);
                                gx[k][t] = MFMA(w2h, bl[k][1], gx[k][t]);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int i = i0 + 8 * k < NMID ? i0 + 8 * k : i0, P = Pk[k];
                    float v[8];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float c = __builtin_fmaf(cx[k][t][e], DOWN, cm[k][t][e]), g = __builtin_fmaf(gx[k][t][e], DOWN, gm[k][t][e]);
                            v[4 * t + e] = c * __builtin_amdgcn_rcpf(1.0f + __expf(-g));
                        }
                    f16x8 h, l;
                    split8(v, h, l);
                    if (edge_t) {          // (a tile on the frame's border only: Y is zero outside the image = k4's zero padding)
                        const int r = P / HC, cc = P - r * HC, y = ty0 - 2 + r, x = tx0 - 2 + cc;
                        if (!(y >= 0 && y < p.H && x >= 0 && x < p.W)) h = l = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    }
                    if (lg < 3 && 16 * i + li < NY) {
                        *(f16x8*)(smem + AH + lg * PL + P * 16) = h;
                        *(f16x8*)(smem + AL + lg * PL + P * 16) = l;
                    }
                }
            };
            static_assert(NMID > 16 && NMID <= 24, "three passes of one tile slot each cover the region");
            pass(std::integral_constant<int, 2>{}, wave);
            if (wave + 16 < NMID) pass(std::integral_constant<int, 1>{}, wave + 16);
        }
        __syncthreads();
        // ---------------- P3: b' = lrelu(k4(Y)); out = conv3(a' | b') + x ----------------
        load_x(j + slots < run_len ? j + slots : j);               // the next tile's x, in flight while this tile finishes (unconditional: a conditional load keeps the OLD x alive through P2)
        {
            f32x4 cm[2][2], cx[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) cm[u][0] = cm[u][1] = cx[u][0] = cx[u][1] = z4;
            conv33_split(smem, AH + pconv_w, w0_w, w1_w, OFF_K4, cm, cx);
            f16x8 bph[2], bpl[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) join_split8<true>(cm[u], cx[u], bph[u], bpl[u]);
            __builtin_amdgcn_sched_barrier(0);
            // the residual x of the wave's pixels in conv3's result layout (a lane: channels 16 lg .. 16 lg + 15 of pixel li)
            f16x8 rh[2][2], rl[2][2];
            int opix[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int y = ty0 + row0 + u, x = tx0 + seg * 16 + li;
                const bool ok = y < p.H && x < p.W;
                const int pix = (n * p.H + y) * p.W + x;
                opix[u] = ok ? pix : -1;
                const int o0 = !ok || lg == 3 ? OOB : (lg < 2 ? pix * 64 + lg * 32 : o_hi8 + pix * 16), o1 = ok && lg < 2 ? o0 + 16 : OOB;
                const int lo_d = lg < 2 ? o_lo32 : o_lo8 - o_hi8;
                rh[u][0] = gload(rs, o0); rh[u][1] = gload(rs, o1);
                rl[u][0] = gload(rs, o0 == OOB ? OOB : o0 + lo_d); rl[u][1] = gload(rs, o1 == OOB ? OOB : o1 + lo_d);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float d[4][4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    f16x8 w3h[2], w3l[2];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int o = OFF_C3 + c3_t(t) + (li * 6 + ks * 3 + lg) * 16;
                        w3h[ks] = lds16(smem, o, li < r4(t) && lg < 3);
                        w3l[ks] = lds16(smem, WLO + o, li < r4(t) && lg < 3);
                    }
                    f32x4 dm = MFMA(w3h[0], aph[u], z4);
                    dm = MFMA(w3h[1], bph[u], dm);
                    f32x4 dc = MFMA(w3l[0], aph[u], z4);
                    dc = MFMA(w3l[1], bph[u], dc);
                    dc = MFMA(w3h[0], apl[u], dc);
                    dc = MFMA(w3h[1], bpl[u], dc);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[t][e] = __builtin_fmaf(dc[e], DOWN, dm[e]);
                }
                if (opix[u] >= 0 && lg < 3) {
                    float o0[8], o1[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o0[e] = d[0][e] + __builtin_fmaf((float)rl[u][0][e], DOWN, (float)rh[u][0][e]);
                        o0[4 + e] = d[1][e] + __builtin_fmaf((float)rl[u][0][4 + e], DOWN, (float)rh[u][0][4 + e]);
                        o1[e] = d[2][e] + __builtin_fmaf((float)rl[u][1][e], DOWN, (float)rh[u][1][e]);
                        o1[4 + e] = d[3][e] + __builtin_fmaf((float)rl[u][1][4 + e], DOWN, (float)rh[u][1][4 + e]);
                    }
                    f16x8 h0, l0, h1, l1;
                    split8(o0, h0, l0);
                    split8(o1, h1, l1);
                    const long pix = opix[u];
                    if (lg == 2) {
                        *(f16x8*)(p.out + (long)o_hi8 + pix * 16) = h0;
                        *(f16x8*)(p.out + (long)o_lo8 + pix * 16) = l0;
                    } else {
                        char* oh = p.out + pix * 64 + lg * 32;
                        *(f16x8*)oh = h0; *(f16x8*)(oh + 16) = h1;
                        *(f16x8*)(oh + o_lo32) = l0; *(f16x8*)(oh + o_lo32 + 16) = l1;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                            // Y is dead
    }
}

// NCHW fp32 (40 channels) <-> split planes.  A thread: one pixel's k-octet (8 channels).
__global__ void split_from_nchw(const float* x, char* planes, int npx, int hw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx * 5) return;
    const int oct = i / npx, pix = i - oct * npx, n = pix / hw, q = pix - n * hw;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = x[((long)n * 40 + oct * 8 + e) * hw + q];
    f16x8 h, l;
    split8(f, h, l);
    const long o = oct < 4 ? (long)pix * 64 + oct * 16 : (long)npx * 64 + (long)pix * 16;
    *(f16x8*)(planes + o) = h;
    *(f16x8*)(planes + (long)npx * 80 + o) = l;
}
__global__ void split_to_nchw(const char* planes, float* x, int npx, int hw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx * 5) return;
    const int oct = i / npx, pix = i - oct * npx, n = pix / hw, q = pix - n * hw;
    const long o = oct < 4 ? (long)pix * 64 + oct * 16 : (long)npx * 64 + (long)pix * 16;
    const f16x8 h = *(const f16x8*)(planes + o), l = *(const f16x8*)(planes + (long)npx * 80 + o);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[((long)n * 40 + oct * 8 + e) * hw + q] = __builtin_fmaf((float)l[e], DOWN, (float)h[e]);
}

}  // namespace

size_t pan_scpa_split_blob_bytes() { return 2 * WLO; }

// the fp16 kernel's blob (pan_scpa_pack) twice: fp16(w) and fp16((w - fp16(w)) * 2^11) (k2's bias, fp32, in the first)
void pan_scpa_split_pack(const float* c1a, const float* c1b, const float* k1, const float* k2, const float* k2b, const float* k3, const float* k4, const float* c3, void* blob) {
    char* w = (char*)blob;
    for (int i = 0; i < 2 * WLO; ++i) w[i] = 0;
    const float* src[7] = {c1a, c1b, k1, k2, k3, k4, c3};
    const int cnt[7] = {20 * 40, 20 * 40, 20 * 20 * 9, 20 * 20, 20 * 20 * 9, 20 * 20 * 9, 40 * 40};
    std::vector<float> hi[7], lo[7];
    for (int i = 0; i < 7; ++i) {
        hi[i].resize(cnt[i]); lo[i].resize(cnt[i]);
        for (int e = 0; e < cnt[i]; ++e) {
            const f16 h = (f16)src[i][e];
            hi[i][e] = (float)h;
            lo[i][e] = (float)(f16)((src[i][e] - (float)h) * UP);
        }
    }
    const std::vector<float> zb(20, 0.f);
    pan_scpa_pack(hi[0].data(), hi[1].data(), hi[2].data(), hi[3].data(), k2b, hi[4].data(), hi[5].data(), hi[6].data(), w);
    pan_scpa_pack(lo[0].data(), lo[1].data(), lo[2].data(), lo[3].data(), zb.data(), lo[4].data(), lo[5].data(), lo[6].data(), w + WLO);
}

bool pan_scpa_split_ok(int N, int H, int W) { return (long)N * H * W * 160 < 0x7fffffffL; }

int pan_split_from_nchw(const float* x, void* planes, int N, int H, int W, hipStream_t s) {
    const int npx = N * H * W;
    hipLaunchKernelGGL(split_from_nchw, dim3((unsigned)(((long)npx * 5 + 255) / 256)), dim3(256), 0, s, x, (char*)planes, npx, H * W);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}
int pan_split_to_nchw(const void* planes, float* x, int N, int H, int W, hipStream_t s) {
    const int npx = N * H * W;
    hipLaunchKernelGGL(split_to_nchw, dim3((unsigned)(((long)npx * 5 + 255) / 256)), dim3(256), 0, s, (const char*)planes, x, npx, H * W);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

int pan_scpa_split_launch(const void* in, void* out, const void* d_blob, int N, int H, int W, hipStream_t s) {
    if (!pan_scpa_split_ok(N, H, W)) return set_error(INNFER_ERR_UNSUPPORTED, "pan_scpa_split: tensor too large for 32-bit buffer offsets");
    int dev = 0, num_cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    static std::atomic<int> cus[64] = {};
    num_cus = cus[dev & 63].load(std::memory_order_relaxed);
    if (!num_cus) {
        if (hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || num_cus <= 0) num_cus = 256;
        cus[dev & 63].store(num_cus, std::memory_order_relaxed);
    }
    static std::atomic<unsigned long long> attr_done{0};
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute((const void*)pan_scpa_split, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
            (void)hipGetLastError();
            return set_error(INNFER_ERR_UNSUPPORTED, "pan_scpa_split: the fused SCPA block needs 149 KB of LDS per workgroup (gfx950)");
        }
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    SplitKP k{};
    k.in = (const char*)in; k.out = (char*)out; k.w = (const char*)d_blob; k.npx = N * H * W; k.N = N; k.H = H; k.W = W;
    k.tiles_x = (W + TW - 1) / TW; k.tiles_y = (H + TH - 1) / TH;
    const long total = (long)N * k.tiles_x * k.tiles_y;
    k.total = (int)total;
    const int grid = total < num_cus ? (int)total : num_cus;
    hipLaunchKernelGGL(pan_scpa_split, dim3(grid), dim3(512), LDS_BYTES, s, k);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

}  // namespace innfer
