// The HR tail of the SR networks as ONE kernel: the last upconv_block -> HR_conv0 (+ act) -> conv_last, chained through LDS (round 6).
//
// Replaces, for RRDBNet (RRDBNet_arch.py:31-42, block.py:348-361):  Upsample(nearest 2x) -> conv3x3(64 -> 64) -> LeakyReLU   (the second upconv_block)
//                                                                     -> HR_conv0 = conv3x3(64 -> 64) -> LeakyReLU -> conv_last = conv3x3(64 -> out_nc)
// The layer-by-layer engine wrote the up-conv's [64 ch, 4H x 4W] result (4.25 GB at 1080p: the launch is bound by that store) and read it back in the fused
// HR_conv0 + conv_last launch.  Here an output tile of 16 x 32 HR pixels never leaves the CU:
//   stage A  the up-conv as the four 2 x 2-tap phases of the equivalent transposed conv (the panels of the one-visit form: conv_pack_up2x_phases, plane row order) on the
//            tile's 18 x 34 neighbourhood (HR_conv0's halo, recomputed: 612 pixels for 512 = +20 % of the up-conv's MFMAs), its activation, fp16 rounding -- the values
//            the unchained engine stored -- written into an LDS image of the tile (hr_slot below), zeros outside the frame (HR_conv0's zero padding);
//   stage B  HR_conv0 on that LDS-resident tile: the walk of conv3x3_pc<2,4,..> (same fragments, same MFMA order per value), weights through a ring in LDS;
//   fuse     conv_last in HR_conv0's epilogue: conv3x3_fuse.h as it is (rim pixels by fuse_combine_kernel).
// Every value sees the same operands in the same order as in the two unchained launches: results are BIT-IDENTICAL to them (tests/test_gpu_parity.py
// test_hr_chain_equals_the_launches_it_replaces, scripts/r6/fuzz_chain.py).
//
// Workgroup: 8 waves (two per SIMD, 256 registers each, no scratch), one persistent workgroup per CU, XCD-aware tile walk as in conv3x3_pc.  No loader waves:
//   * stage A's weights never touch LDS.  Wave w owns phase w / 2 and one 32-channel half of its outputs for all of the phase's 153 virtual pixels (9 x 17 LR-grid
//     positions, linearised into 10 MFMA column groups -- no 17-wide rows padded to 32); its sixteen A fragments (2 input groups x 4 taps x 2 channel tiles, 64 registers:
//     each one contiguous 1 KB of the panel) are loaded ONCE per kernel, so stage A has no weight traffic and needs no barrier (the 128 KB of phase panels through a
//     16-KB LDS ring would be a barrier every 640 cycles; reloading them per tile from L2 cost 0.38 of the first version's 4.1 ms);
//   * the LR input tile (10 x 18 pixels x 64 channels, rows padded to 25 pixels so that the linearised pixel walk stays bank-conflict free), stage B's weight pieces
//     (one tap COLUMN of one input group = 12 KB, three ring slots) and conv_last's fragments arrive by LDS-DMA issued by the consumer waves themselves.
// LDS: LR tile 2 x 16 KB | HR tile 2 x 41 KB | weight ring 3 x 12 KB | conv_last fragments 4 KB | biases = 155 KB.
// Measured, ablated and bounded in DESIGN.md 3.2b / docs/EXPERIMENTS.md 103-111; diagnostic driver scripts/r6/hr_chain_micro.cpp (-DINNFER_ABLATE).
#include "common.h"

#include <type_traits>

#define FP32_VALUE(x) asm("" : "+v"(x))
#ifndef INNFER_IN_AUX
#define INNFER_IN_AUX 0
#endif
#pragma clang fp contract(off)

namespace innfer {

namespace {

#define STAMP(i) do { } while (0)
#include "conv3x3_kp.h"
#include "conv3x3_fuse.h"
#ifdef INNFER_ABLATE
#define CH_ABL(bit) (p.abl & (bit))
#else
#define CH_ABL(bit) false
#endif

struct ChainP {
    KP kp;                        // what fused_last_epilogue reads: H, W (the HR grid), act (HR_conv0's), fl_*, out_denorm, out_round16
    const f16* in; long in_img_stride, in_gbytes;      // the up-conv's input slab: two 32-channel groups of [N, h, w] pixels
    int h, w;                     // its grid; the HR grid is 2h x 2w
    const f16* wup; const float* bup;      // conv_pack_up2x_phases(.., rowp 1) panels [phase][group][tap][64][64 B] and the bias once per phase (256 floats)
    const f16* whr; const float* bhr;                  // HR_conv0: conv_pack(64, 64, rowp 1) [group][tap][64][64 B], 64 biases
    int N, tiles_x, tiles_y, total, rev;
#ifdef INNFER_ABLATE
    int abl;                      // diagnostic build only (scripts/r6/hr_chain_micro.cpp): 1 no stage A reads / MFMAs, 2 no HR-tile stores, 4 no stage B reads / MFMAs, 8 no fused epilogue, 16 no LR-tile DMA, 32 no stage B weight DMA, 64 no stage A weight loads
#endif
};

constexpr int CH_IN_BYTES = (((16 + 2) * LWP + 15) / 16) * 1024;       // one 32-channel group of the 18 x 36-pixel HR tile, as conv3x3_pc stages it (41 984)
constexpr int CH_LRP = 25;                                               // LR tile row pitch (pixels): 17 + 8 -- see stage A's per-lane constants
constexpr int CH_LR_CG = 16 * 1024;                                      // 10 rows x 25 px x 64 B = 16 000 -> 16 DMA pieces
constexpr int CH_LRT = 0;                                                // (first: its addresses stay below 64 KB, the input group is an immediate of the read)
constexpr int CH_HRT = CH_LRT + 2 * CH_LR_CG;
constexpr int CH_WB = CH_HRT + 2 * CH_IN_BYTES;
constexpr int CH_WB_SLOT = 3 * 64 * 64;                                  // one tap column (3 taps) x 64 rows x 64 B
constexpr int CH_FLW = CH_WB + 3 * CH_WB_SLOT;
constexpr int CH_BUP = CH_FLW + 4096;
constexpr int CH_BHR = CH_BUP + 1024;
constexpr int CH_LDS = CH_BHR + 256;
static_assert(CH_LDS <= 160 * 1024, "the chain's stages must fit the CU's LDS");
static_assert(2 * CH_IN_BYTES >= 27 * FUSE_PITCH * 4, "the fused last conv parks its 27 product planes in the HR tile");

// HR tile image in LDS: rows of 36 pixels x 64 B per 32-channel group as in conv3x3_pc, but inside a row the EVEN columns come first and the odd ones 72 slots behind,
// and the four 16-byte octets of pixel pair j rotate by j: slot(col, octet) = 4 (col / 2) + ((octet + col / 2) & 3) + 72 (col & 1).  Stage B's fragment reads (16 consecutive
// columns from any start, one octet per 16-lane half) stay bank-conflict free, and stage A's stores -- 8 lanes holding every other column (one phase) -- conflict two ways
// instead of four as in conv3x3_pc's pixel-major image, where a phase's columns all fall into the same 64-byte half of every 128-byte bank window (ablation: the stores
// were 0.7 of the first version's 4.1 ms).
__host__ __device__ constexpr int hr_slot(int col, int oct) { return 4 * (col >> 1) + ((oct + (col >> 1)) & 3) + 72 * (col & 1); }

struct ChGrp { int w, lr1, lr2, hrA1, hrA2, c1, c2; };
// pixel group g of stage A: lanes li < w are in lattice row k_g, column m0 + li; the others in row k_g + 1, column m0 + li - 17.
//   LR-tile pixel (tap 0, 0), unswizzled byte offset  = li * 64 + lg * 16 + (li < w ? lr1 : lr2)
//   HR-tile pixel, byte offset (hr_slot image)        = (ra * 36 * 64 + cb * 72 * 16) + li * 64 + 16 ((lg + li + (li < w ? c1 : c2)) & 3) + (li < w ? hrA1 : hrA2)
constexpr ChGrp ch_grp(int g) {
    const int v0 = 16 * g, k = v0 / 17, m0 = v0 - 17 * k;
    return ChGrp{17 - m0, (k * 25 + m0) * 64, ((k + 1) * 25 + m0 - 17) * 64, 2 * k * 36 * 64 + 64 * m0, 2 * (k + 1) * 36 * 64 + 64 * (m0 - 17), m0 & 3, (m0 - 17) & 3};
}

// ACT_UP: the up-conv's activation (1 LeakyReLU(0.2), 2 ReLU) -- compile-time: a run-time choice triples the unrolled epilogue of stage A
template <int ACT_UP>
__global__ __launch_bounds__(512, 1) void hr_chain_kernel(const ChainP p) {
    constexpr int RPW = 2, NT = 4, MT = 2 * RPW;
    constexpr int OOB = (int)0x80000000;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;

    // ---- workgroup -> tiles: as conv3x3_pc (blocks b, b + 8, .. share an XCD and walk a contiguous run of the tile list) ----
    const int bid = blockIdx.x, xcd = bid & 7;
    const int run_q = p.total >> 3, run_r = p.total & 7;
    const int run_start = xcd < run_r ? xcd * (run_q + 1) : run_r * (run_q + 1) + (xcd - run_r) * run_q;
    const int run_len = run_q + (xcd < run_r ? 1 : 0);
    const int slots = ((int)gridDim.x + 7 - xcd) >> 3;
    const int per_img = p.tiles_x * p.tiles_y;
    const int j0 = bid >> 3;
    if (j0 >= run_len) return;
    auto decode = [&](int jj, int& lid_, int& n_, int& ty0_, int& tx0_) __attribute__((always_inline)) {
        lid_ = run_start + (p.rev ? run_len - 1 - jj : jj);
        n_ = lid_ / per_img;
        const int tile = lid_ - n_ * per_img;
        const int ty = tile / p.tiles_x;
        ty0_ = ty * 16;
        tx0_ = (tile - ty * p.tiles_x) * TW;
    };

    // ---- stage A, per-lane constants ----
    // Wave w computes phase ph = w / 2 = 2a + b of the transposed conv (output pixel (2y + a, 2x + b) reads LR rows y + a - 1, y + a and columns x + b - 1, x + b)
    // for one 32-channel half of the outputs, on the tile's 18 x 34 neighbourhood (local HR rows r = 0 .. 17 <-> frame row ty0 - 1 + r, ty0 even: r even is a = 1).
    // Local HR pixel (r, c) = (2k + (1 - a), 2m + (1 - b)), k = 0 .. 8, m = 0 .. 16, reads LR-tile pixels (k + dr, m + dc), dr, dc in {0, 1} (LR-tile origin = LR pixel
    // (ty0 / 2 - 1, tx0 / 2 - 1)) with the phase panel's tap 2 dr + dc -- the SAME for all four phases.  Virtual pixel v = 17 k + m = 0 .. 152; group g (0 .. 9) holds
    // v = 16 g + li.  LR tile rows are 25 pixels apart, so pixel index P = 25 k + m = v + 8 k: the 16 lanes of a fragment read walk consecutive pixels with
    // at most one jump of 8 -- P mod 8 stays a permutation inside each half of a ds_read_b128 lane group, i.e. the 16-byte-slot XOR swizzle of conv3x3_pc (slot ^=
    // 2 bit2(P)) keeps every read bank-conflict free (a pitch of 18 would conflict two ways on every row change).
    const int ph = wave >> 1, th = wave & 1;          // th: the wave's half of the 64 output channels = slab plane th (channel tiles 2 th, 2 th + 1 of the plane row order)
    const int ra = 1 - (ph >> 1), cb = 1 - (ph & 1);
    // Group g holds virtual pixels v = 16 g + li: rows k_g (lanes li < w_g) and k_g + 1 (the others) of the 9 x 17 lattice -- the per-lane offsets of its LR-tile pixel and
    // of the HR-tile pixel it produces are a lane term plus one of two COMPILE-TIME constants per group (ch_grp above): a compare and a select where ten tile-invariant
    // registers were held (and spilled: the kernel lives at its register budget).  Lanes li >= 9 of group 9 (v >= 153) have no pixel: they read inside the LDS
    // allocation, compute into their own MFMA columns and store nothing.
    // The phase's A fragments straight from the panel, ONCE per kernel: the wave's phase and channel half never change, so its sixteen fragments (2 input groups x 4 taps x
    // 2 channel tiles, 64 registers) serve every tile -- no weight traffic at all in stage A.  Fragment (group cg, tap rank, tile t) = 1 KB at cg * 16 KB + (rank * 64 + 16 t) * 64,
    // lane (li, lg) reads row li, octet lg (buffer loads: one per-lane offset register and a scalar offset per fragment).
    const int wa_voff = li * 64 + ((lg ^ (((li >> 2) & 1) << 1)) << 4);
    f16x8 wa[8][2];
#if defined(__HIP_DEVICE_COMPILE__)
    if (!CH_ABL(64)) {
        const __amdgpu_buffer_rsrc_t rwa = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.wup + (long)ph * 2 * 16384 + th * 2048), 0, 2 * 16384, 0x00020000);
#pragma unroll
        for (int sr = 0; sr < 8; ++sr)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                wa[sr][t] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rwa, wa_voff, (sr >> 2) * 16384 + ((sr & 3) * 64 + t * 16) * 64, 0));
    }
#else
    (void)wa_voff;
#endif

    // ---- stage B, per-lane constants (conv3x3_pc's consumers: wave cw owns tile rows 2 cw, 2 cw + 1) ----
    const int cw = wave;
    int boffs[3][2];              // byte offset of the wave's first halo row, column 16 seg + li + s, octet lg: [tap column s][segment]
#pragma unroll
    for (int sc = 0; sc < 3; ++sc)
#pragma unroll
        for (int seg = 0; seg < 2; ++seg) boffs[sc][seg] = CH_HRT + cw * RPW * (LWP * 64) + hr_slot(16 * seg + li + sc, lg) * 16;
    const int aoffs = li * 64 + ((lg ^ (((li >> 2) & 1) << 1)) << 4);

    // ---- LDS-DMA issue (every wave its share) ----
    // LR tile: 32 pieces (group cg = q / 16, pixels 16 (q % 16) .. + 15); wave w issues q = w, w + 8, w + 16, w + 24: two offsets
    int lroff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int P = 16 * (wave + 8 * i) + (lane >> 2);
        const int row = P / CH_LRP, col = P - row * CH_LRP;
        const int slot = (lane & 3) ^ (((P >> 2) & 1) << 1);
        lroff[i] = (row < 10 && col < 18) ? ((row * p.w + col) * 32 + slot * 8) * 2 : OOB;
    }
    auto issue_lr = [&](int n, int ty0, int tx0) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (CH_ABL(16)) return;
        const int Y = (ty0 >> 1) - 1, X = (tx0 >> 1) - 1;
        const char* base = (const char*)(p.in + (long)n * p.in_img_stride) + ((long)Y * p.w + X) * 64;
        int vo[2] = {lroff[0], lroff[1]};
        if (Y < 0 || Y + 10 > p.h || X < 0 || X + 18 > p.w) {
            int ln = lane;
            asm volatile("" : "+v"(ln));          // (border tiles only: re-derived from an opaque copy -- shared with lroff's set-up the rows / columns are held, i.e. spilled, across the kernel)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int P = 16 * (wave + 8 * i) + (ln >> 2);
                const int row = P / CH_LRP, col = P - row * CH_LRP;
                if (Y + row < 0 || Y + row >= p.h || X + col < 0 || X + col >= p.w) vo[i] = OOB;
            }
        }
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
            const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)(base + cg * p.in_gbytes), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ri, (__attribute__((address_space(3))) void*)(smem + CH_LRT + cg * CH_LR_CG + (wave + 8 * i) * 1024), 16, vo[i], 0, 0, INNFER_IN_AUX);
        }
#else
        (void)n; (void)ty0; (void)tx0; (void)lroff;
#endif
    };
    // stage B weights, step n = (group n / 3, tap column n % 3): 12 pieces of 1 KB -- piece j: tap row j / 4, quarter j % 4 of its 4 KB; waves 0 .. 7 issue j = w, waves 0 .. 3 also j = w + 8
    auto issue_wb = [&](int n) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (CH_ABL(32)) return;
        const int cg = n / 3, sc = n - 3 * cg;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.whr, 0, 2 * 9 * 4096, 0x00020000);
        char* slot = smem + CH_WB + (n % 3) * CH_WB_SLOT;
        {
            const int j = wave;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(slot + j * 1024), 16, lane * 16, ((cg * 9 + (j >> 2) * 3 + sc) * 4 + (j & 3)) * 1024, 0, 0);
        }
        if (wave < 4) {
            const int j = wave + 8;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(slot + j * 1024), 16, lane * 16, ((cg * 9 + (j >> 2) * 3 + sc) * 4 + (j & 3)) * 1024, 0, 0);
        }
#else
        (void)n;
#endif
    };
    auto wait_vm_all = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    // ---- prologue: conv_last's fragments, the biases, the first tile's LR tile and stage B pieces 0, 1, the first weights of stage A ----
    int lid, n, ty0, tx0;
    decode(j0, lid, n, ty0, tx0);
#if defined(__HIP_DEVICE_COMPILE__)
    if (wave < 4) {
        const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void*)p.kp.fl_w, 0, 4096, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rf, (__attribute__((address_space(3))) void*)(smem + CH_FLW + wave * 1024), 16, lane * 16, wave * 1024, 0, 0);
    }
#endif
    if (wave == 4) *(f32x4*)(smem + CH_BUP + lane * 16) = *(const f32x4*)(p.bup + 4 * lane);
    if (wave == 5 && lane < 16) *(f32x4*)(smem + CH_BHR + lane * 16) = *(const f32x4*)(p.bhr + 4 * lane);
    issue_lr(n, ty0, tx0);
    issue_wb(0);
    issue_wb(1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");

    for (int j = j0; j < run_len; j += slots) {
        const bool has_next = j + slots < run_len;
        // ================================ stage A: the up-conv's four phases on the 18 x 34 neighbourhood ================================
        asm volatile("s_barrier" ::: "memory");          // B1: every wave has left the previous tile's fused epilogue (its products lie in the HR tile)
        {
            // Pixel group by pixel group: 8 fragment reads (input group x tap) feed 16 MFMAs on the group's two accumulators, then its activation, fp16 rounding -- the
            // values the unchained up-conv stored --, zeros outside the frame, and ONE 16-byte store into the HR tile (a lane's 8 channels are octet lg of its pixel in the
            // wave's 32-channel group: plane row order); the epilogue of group g is placed inside the MFMAs of group g + 1.  Taps in the order of the one-visit kernel's walk
            // (tap column first: ranks 0, 2, 1, 3), input group 0 before 1: the order in which the unchained launch accumulated every value.  The read address is derived
            // from the group's pixel index as the read is issued (4 VALU instructions beside 2 MFMAs): hoisted out of the tile loop they would be forty registers.
            constexpr int NRD = 80, RING = 5, AHEAD = 4;
            const bool edge = ty0 == 0 || ty0 + 16 >= p.kp.H || tx0 == 0 || tx0 + TW >= p.kp.W;
            const float* bl = (const float*)(smem + CH_BUP) + ph * 64 + 32 * th + 8 * lg;
            const f32x4 bias0 = *(const f32x4*)bl, bias1 = *(const f32x4*)(bl + 4);
            f16x8 bq[RING];
            f32x4 acc[2][2];          // [group parity][channel tile]
            // Read addresses: pixel P + tap, octet lg at byte (P + tap) * 64 + 16 (lg ^ 2 bit2(P + tap)) -- with w = the unswizzled offset (bit 8 of w = bit2 of the pixel,
            // bits 4, 5 = lg) that is w ^ ((w >> 3) & 32): three VALU instructions per address, four addresses per pixel group (input group 1 = + 16 KB, an immediate).
            // The tap terms pass through an opaque register once per tile so that the forty sums are not hoisted out of the tile loop (forty registers).
            int li_t = li, lane64 = li * 64 + lg * 16, sum_t = li + lg;          // (opaque per tile, like the tap terms)
            asm volatile("" : "+v"(li_t), "+v"(lane64), "+v"(sum_t));
            const int hr0 = CH_HRT + th * CH_IN_BYTES + ra * (LWP * 64) + cb * (72 * 16) + li_t * 64;
            int tap64[4];
#pragma unroll
            for (int rk = 0; rk < 4; ++rk) { tap64[rk] = ((rk >> 1) * CH_LRP + (rk & 1)) * 64; asm volatile("" : "+v"(tap64[rk])); }
            int ad[4];
            auto ldb = [&](int i) __attribute__((always_inline)) {
                const int g = i >> 3, sr = i & 7, cg = sr >> 2, ti = sr & 3, rank = ((ti & 1) << 1) | (ti >> 1);
                if (cg == 0) {
                    const ChGrp G = ch_grp(g);          // (g is a constant of the unrolled loop)
                    const int w = lane64 + (li_t < G.w ? G.lr1 : G.lr2) + tap64[rank];
                    ad[ti] = w ^ ((w >> 3) & 32);
                }
                return *(const f16x8*)(smem + CH_LRT + cg * CH_LR_CG + ad[ti]);
            };
            auto store_group = [&](int g) __attribute__((always_inline)) {
                f16x8 h;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) {
                        float f = acc[g & 1][t][jx];
                        // (LeakyReLU as max(f, 0.2 f) == f > 0 ? f : 0.2 f for every finite f.  As v_med3_f32(f, 0.2 f, top) with a FINITE top: the builtin max is the IEEE
                        //  maxnum, which canonicalises its operand first -- three instructions per value like compare + select, and this stage is bound by vector issue --
                        //  and so is what LLVM makes of a median whose third operand is +inf (the form this kernel shipped with until the end of round 6: 576 v_max for 288
                        //  values); an inline-asm v_max_f32 is invisible to the hazard recogniser -- as the first reader of an MFMA result it read stale registers)
                        if (ACT_UP == 1) f = __builtin_amdgcn_fmed3f(f, 0.2f * f, 3.0e38f);
                        else if (ACT_UP == 2) f = __builtin_amdgcn_fmed3f(f, 0.f, 3.0e38f);
                        FP32_VALUE(f);
                        h[4 * t + jx] = (f16)f;
                    }
                if (edge) {               // a tile on the frame's border: the neighbourhood's pixels outside the frame are HR_conv0's zero padding (selects, one uniform branch)
                    int v = li;
                    asm volatile("" : "+v"(v));          // (row / column re-derived here: hoisted, the twenty of them would be twenty registers)
                    v += 16 * g;
                    const int k = (v * 241) >> 12, m = v - 17 * k;
                    const int y = ty0 - 1 + 2 * k + ra, x = tx0 - 1 + 2 * m + cb;
                    const bool zero = y < 0 || y >= p.kp.H || x < 0 || x >= p.kp.W;
                    u32x4 hv = __builtin_bit_cast(u32x4, h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[e] = zero ? 0u : hv[e];
                    h = __builtin_bit_cast(f16x8, hv);
                }
                if (CH_ABL(2)) return;
                const ChGrp G = ch_grp(g);
                const bool first = li_t < G.w;
                const int O = hr0 + (first ? G.hrA1 : G.hrA2) + (((sum_t + (first ? G.c1 : G.c2)) & 3) << 4);
                if (g < 9 || first)          // (only the last group has lanes without a pixel: its second row, v >= 153)
                    *(f16x8*)(smem + O) = h;
            };
            if (!CH_ABL(1)) {
#pragma unroll
            for (int i = 0; i < AHEAD; ++i) bq[i] = ldb(i);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 10; ++g) {
#pragma unroll
                for (int sr = 0; sr < 8; ++sr) {
                    const int i = 8 * g + sr, ti = sr & 3, wr = (sr & 4) | ((ti & 1) << 1) | (ti >> 1);          // wr: (input group, tap rank) of step sr
                    if (i + AHEAD < NRD) bq[(i + AHEAD) % RING] = ldb(i + AHEAD);
                    // (a group's first MFMAs take the bias as their C operand: no moves)
                    acc[g & 1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[wr][0], bq[i % RING], sr == 0 ? bias0 : acc[g & 1][0], 0, 0, 0);
                    acc[g & 1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[wr][1], bq[i % RING], sr == 0 ? bias1 : acc[g & 1][1], 0, 0, 0);
                    if (sr == 3 && g > 0) store_group(g - 1);          // (in the shadow of this group's MFMAs)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            store_group(9);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // (vmcnt: this wave's piece(s) of stage B's step 1, issued behind the previous tile's epilogue)
        asm volatile("s_barrier" ::: "memory");          // B2: the HR tile is complete; the LR tile is free
        issue_wb(2);
        int lid2 = 0, n2 = 0, ty2 = 0, tx2 = 0;
        if (has_next) decode(j + slots, lid2, n2, ty2, tx2);

        // ================================ stage B: HR_conv0 on the LDS-resident tile ================================
        // conv3x3_pc's nine-tap walk, one TAP per sub-step u = 3 * step + kernel row r (step = (input group, tap column): one 12-KB ring slot): the tap's four weight fragments
        // against the 2 x 2 pixel fragments (output rows 0, 1 of the wave = halo rows r, r + 1; two segments) it meets -- 16 MFMAs.  Per output value the taps arrive in
        // conv3x3_pc's order (column by column, kernel rows ascending).  Two fragment sets of four (the next tap's load under this tap's MFMAs) instead of conv3x3_pc's
        // twelve live fragments: 16 registers that stage A's resident weights need (20 % more pixel-fragment reads; LDS has the room).
        f16x8 a[2][NT];
        auto lda = [&](int u) __attribute__((always_inline)) {          // tap u -> set u & 1; ring slot (u / 3) % 3
            const int st = u / 3, r = u - 3 * st;
#pragma unroll
            for (int t = 0; t < NT; ++t) a[u & 1][t] = *(const f16x8*)(smem + CH_WB + (st % 3) * CH_WB_SLOT + aoffs + (r * 64 + t * 16) * 64);
        };
        lda(0);
        f32x4 acc2[NT][MT];
        {
            const float* bl = (const float*)(smem + CH_BHR) + 8 * lg;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 bt = *(const f32x4*)(bl + 32 * (t >> 1) + 4 * (t & 1));
#pragma unroll
                for (int m = 0; m < MT; ++m) acc2[t][m] = bt;
            }
        }
        {
            f16x8 bq[3];
            auto ldb = [&](int i) __attribute__((always_inline)) {          // i = 4 u + 2 (output row) + segment
                const int u = i >> 2, st = u / 3, r = u - 3 * st, cg = st / 3, sc = st - 3 * cg, rr = r + ((i >> 1) & 1), seg = i & 1;
                return *(const f16x8*)(smem + cg * CH_IN_BYTES + boffs[sc][seg] + rr * (LWP * 64));
            };
            if (!CH_ABL(4)) { bq[0] = ldb(0); bq[1] = ldb(1); }
#pragma unroll
            for (int st = 0; st < 6; ++st) {
                if (!CH_ABL(4))
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int u = 3 * st + r;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = 4 * u + q, m = q;                       // m = 2 (output row) + segment
                        if (i + 2 < 72) bq[(i + 2) % 3] = ldb(i + 2);
                        if (q == 0 && u + 1 < 18) lda(u + 1);                 // (step st + 1's slot is visible since the barrier before step st)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc2[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 1][t], bq[i % 3], acc2[t][m], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (st < 4) {
                    // end of step st: the piece(s) of step st + 2 this wave issued a step ago have landed (step 3: the next tile's LR pieces, issued after them, stay in flight)
                    if (st == 3 && has_next) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else wait_vm_all();
                    asm volatile("s_barrier" ::: "memory");          // steps st + 1, st + 2 are visible; the slots of steps <= st are free
                    if (st + 3 < 6) issue_wb(st + 3);
                    // the next tile's LR tile behind the last weight piece: the only traffic here that may come from HBM gets steps 3 .. 5 and the epilogue's first half to land,
                    // and no weight wait ever waits for it (issued behind B2 it was waited for with step 3's piece: 0.2 of 3.9 ms in the ablation)
                    if (st == 2 && has_next) issue_lr(n2, ty2, tx2);
                    // after step 3 the ring's slot 0 (step 3's: its fragments are in registers by now) takes the NEXT tile's step 0 -- the panels do not depend on the
                    // tile; steps 4 and 5 run without a barrier (slot 1 = step 4's is refilled behind the epilogue's barriers, slot 2 = step 5's behind B2)
                    if (st == 3 && has_next) issue_wb(0);
                }
            }
        }
        // ================================ conv_last in the epilogue (conv3x3_fuse.h) ================================
        {
            // (an opaque per-tile copy of the lane index: otherwise the epilogue's per-lane addresses, ring indices and predicates -- all tile-invariant -- are hoisted
            //  out of the tile loop and held across stages A and B, which then spill: ~100 registers)
            int lane_t = lane;
            asm volatile("" : "+v"(lane_t));
            asm volatile("s_barrier" ::: "memory");          // every wave has read its last fragments of the HR tile: it may be overwritten
            if (!CH_ABL(8)) fused_last_products<RPW, NT>(p.kp, acc2, smem + CH_HRT, smem + CH_FLW, cw, lane_t);
            // the next tile's LR tile and first weight piece have landed (before the sums' global stores join the queue): visible to every wave behind the barrier
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");          // the tile's products are in LDS
            if (!CH_ABL(8)) fused_last_sums<RPW>(p.kp, smem + CH_HRT, n, ty0, tx0, cw, lane_t, lid);
        }
        if (has_next) issue_wb(1);                       // (every wave is past step 4: the epilogue's barriers; it lands under the next tile's stage A and is waited for before B2)
        lid = lid2; n = n2; ty0 = ty2; tx0 = tx2;
    }
}

}  // namespace

// L: the HR_conv0 launch with the fused last conv as net.hip builds it (fuse_* set, H x W the HR grid, in unused); the up-conv's operands beside it.
int hr_chain_launch(const ConvLaunch& L, const f16* up_in, long up_in_gstride, const f16* up_wpk, const float* up_bias, int up_act, hipStream_t s) {
    double gt_flops = 0, gt_bytes = 0;
    if (gt_on()) {
        const double px = (double)L.N * L.H * L.W;
        gt_flops = 2.0 * 9.0 * (64.0 * 64 + 64.0 * 64 + 64.0 * L.fuse_oc) * px;
        gt_bytes = px / 4 * 128.0 + px * L.fuse_oc * (L.fuse_out_mode == 2 ? 1.0 : L.fuse_out_mode == 1 ? 4.0 : 2.0) + 9.0 * (2 * 64.0 * 64 + 64.0 * L.fuse_oc) * 2.0;
    }
    GtScope gt(s, "hr_chain: upconv -> HR_conv0 -> conv_last", gt_flops, gt_bytes);
    if (!hr_chain_ok(L) || (up_act != 1 && up_act != 2)) return set_error(INNFER_ERR_UNSUPPORTED, "hr_chain: 64 -> 64 -> 64 -> <= 3 channels on whole 16 x 32 HR tiles, act 1 / 2");
    static int attr_dev_mask = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!(attr_dev_mask & (1 << (dev & 31)))) {
        INNFER_HIP(hipFuncSetAttribute((const void*)hr_chain_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS));
        INNFER_HIP(hipFuncSetAttribute((const void*)hr_chain_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS));
        attr_dev_mask |= 1 << (dev & 31);
    }
    ChainP c{};
    c.kp.H = L.H; c.kp.W = L.W; c.kp.act = L.act;
    c.kp.fl_w = L.fuse_w; c.kp.fl_bias = L.fuse_bias; c.kp.fl_side = L.fuse_side; c.kp.fl_out = L.fuse_out; c.kp.fl_oc = L.fuse_oc; c.kp.fl_out_mode = L.fuse_out_mode;
    c.kp.out_denorm = L.out_denorm; c.kp.out_round16 = L.out_round16;
    c.h = L.H / 2; c.w = L.W / 2;
    c.in = up_in; c.in_gbytes = up_in_gstride * 2; c.in_img_stride = (long)c.h * c.w * 32;
    c.wup = up_wpk; c.bup = up_bias;
    c.whr = L.wpk; c.bhr = L.bias;
    c.N = L.N; c.tiles_x = L.W / TW; c.tiles_y = L.H / 16;
    const long total = (long)L.N * c.tiles_x * c.tiles_y;
    c.total = (int)total; c.rev = L.rev ? 1 : 0;
#ifdef INNFER_ABLATE
    c.abl = getenv("INNFER_ABL") ? atoi(getenv("INNFER_ABL")) : 0;
#endif
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const long grid = total < cus ? total : cus;
    if (up_act == 1) hipLaunchKernelGGL(hr_chain_kernel<1>, dim3((unsigned)grid), dim3(512), CH_LDS, s, c);
    else hipLaunchKernelGGL(hr_chain_kernel<2>, dim3((unsigned)grid), dim3(512), CH_LDS, s, c);
    INNFER_HIP(hipGetLastError());
    const long nthr = (long)L.N * (L.H / 16) * (L.W / 32) * 92;
    hipLaunchKernelGGL(fuse_combine_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, (const float*)L.fuse_side, L.fuse_bias, L.fuse_out, L.fuse_out_mode, L.out_denorm, L.out_round16, L.fuse_oc,
                       L.N, L.H, L.W);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

bool hr_chain_ok(const ConvLaunch& L) {
    return conv_fuse_last_ok(L) && L.K == 64 && L.C == 64 && L.rowp == 1 && (L.act == 1 || L.act == 2 || L.act == 0) && L.H % 16 == 0 && L.W % 32 == 0 && L.W / 2 > 16 &&
           (long)(L.H / 2) * (L.W / 2) * 64 < 0x7fffffffL && (long)L.N * (L.H / 16) * (L.W / 32) < 0x7fffffffL;
}

}  // namespace innfer
