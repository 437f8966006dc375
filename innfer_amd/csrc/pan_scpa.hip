// One SCPA block of PAN (reference architectures/PAN_arch.py:58-105) as ONE launch: x -> x + conv3(cat[lrelu(k1(lrelu(conv1_a x))),
// lrelu(k4(k3(b) * sigmoid(k2(b) + bias)))]), b = lrelu(conv1_b x), without leaving the CU.
//
// Until round 4 a block was five launches of the SR path's halo-tile kernel with four HBM round trips of 32-channel-padded slabs (16 slab
// groups moved per block for 4 that carry the block's input and output); at 20 / 40 channels those launches are bound by their loads and
// stores and by launch latency, never by the matrix pipe (DESIGN: PAN).  Here a persistent workgroup (8 waves, one per CU, the whole LDS)
// owns a 16 x 32 pixel tile:
//   X   (20 x 36 halo pixels x 40 channels, 80 B per pixel)  global -> LDS by LDS-DMA, out-of-image pixels zero-filled by the buffer range
//       check (= the zero padding of every conv of the block: conv1_a/b have no bias, so A and B are zero wherever x is)
//   P1  A | B = lrelu(conv1_a | conv1_b (x)) on the whole halo tile        -> LDS (20 real channels in 48 B per pixel)
//   P2a a' = lrelu(k1(A)) on the tile's 512 pixels                         -> registers: the MFMA result of a 32-row panel whose rows are permuted
//       like the SR kernel's (a lane ends with 8 consecutive channels of its pixel) IS the B operand of the 1x1 conv3
//   P2b Y = k3(B) * sigmoid(k2(B) + bias) on the 18 x 36 middle region, zero outside the image (k4's zero padding)   -> LDS (over A)
//   P3  b' = lrelu(k4(Y)); out = conv3(a' | b') + x  (x from the X tile, held in registers since P1)               -> global slab
// The block's weights (28.8 KB of fp16 values, 35 KB as MFMA A fragments without their all-zero rows / k-octets, pan_scpa_layout.h) stay in LDS for the whole
// launch; the next tile's X is fetched while P2 / P3 run.  Arithmetic per value as in the five-launch schedule: fp16 operands, fp32
// accumulation, A / B / Y / a' / b' rounded to fp16 where that schedule stored them -- the same roundings, different summation order of the
// MFMA k-steps only where a conv's taps are walked in another order (none: taps in (dy, dx) order, one k-step per tap).
#include <atomic>
#include <type_traits>
#include "common.h"
#include "pan_scpa_layout.h"

namespace innfer {

namespace {

using namespace scpa;                                   // the blob layout (pan_scpa_layout.h)

struct ScpaKP {
    const f16* in; f16* out; long G;        // slabs of two 32-channel groups (40 real channels), group stride G elements
    const char* w;                          // the block's blob (W_BYTES)
    int N, H, W, tiles_x, tiles_y, total;
    int in_c8, out_c8;                      // != 0: channels 32..39 of the input / output slab travel COMPACT -- 16 bytes per pixel at (G elements + pixel * 8) instead of the first 16 bytes
                                            // of a 64-byte group-1 pixel (whose other 48 bytes are zeros no SCPA block reads): between two SCPA blocks of the trunk (round 5)
    int abl;                                // diagnostic build only (make ablate, INNFER_SCPA_ABL): skip 1 P1, 2 P2a, 4 P2b, 8 P3's MFMAs, 16 the X fetch, 32 the stores
#ifdef INNFER_STAMPS
    unsigned long long* stamps;             // diagnostic build only (scripts/r6/scpa_micro.cpp): per workgroup, wave 0's shader-clock cycles in P1 / P2a / P2b / P3 (barriers included), its tile count, and the shader-clock / 100-MHz cycles of its tile loop (their quotient x 100 MHz = the clock the chip held)
#endif
};
// in-kernel phase stamps of the diagnostic build (s_memtime = shader-clock cycles); nothing in the shipped library
#ifdef INNFER_STAMPS
unsigned long long* g_scpa_stamps = nullptr;
#define ST_DECL unsigned long long st_t = 0, st_acc[5] = {0, 0, 0, 0, 0}; const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#define ST_BEGIN st_t = __builtin_amdgcn_s_memtime(); st_acc[4] += 1;
#define ST_MARK(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[i] += n_ - st_t; st_t = n_; }
#define ST_END if (p.stamps && threadIdx.x == 0) { for (int i_ = 0; i_ < 5; ++i_) p.stamps[blockIdx.x * 7 + i_] = st_acc[i_]; p.stamps[blockIdx.x * 7 + 5] = __builtin_amdgcn_s_memtime() - st_c0; p.stamps[blockIdx.x * 7 + 6] = __builtin_amdgcn_s_memrealtime() - st_r0; }
#else
#define ST_DECL
#define ST_BEGIN
#define ST_MARK(i)
#define ST_END
#endif

// (LeakyReLU as v_med3_f32(x, 0.2 x, top) == x > 0 ? x : 0.2 x for every |x| <= top: a packed multiply and ONE instruction per value instead of multiply + compare + select -- round 6:
//  the block is bound by vector issue, 5.5 VALU instructions per MFMA by the counters, profiles/r6/pmc_pan.txt.  `top` is a finite constant on purpose: with +inf LLVM rewrites the median
//  as max(x, 0.2 x) and, in IEEE mode, puts a canonicalising v_max x, x in front of it -- two instructions per value, 224 of the tile's 1467)
constexpr float LRELU_TOP = 3.0e38f;
__device__ __forceinline__ f16x8 lrelu8(const f32x4& a, const f32x4& b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f16x8 v;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4& x = h ? b : a;
            const f2 s = f2{x[2 * q], x[2 * q + 1]} * f2{0.2f, 0.2f};               // (two values per v_pk_mul_f32)
            v[4 * h + 2 * q] = (f16)__builtin_amdgcn_fmed3f(x[2 * q], s[0], LRELU_TOP);
            v[4 * h + 2 * q + 1] = (f16)__builtin_amdgcn_fmed3f(x[2 * q + 1], s[1], LRELU_TOP);
        }
    return v;
}

// The PA gate of a lane's eight channels: c * sigmoid(g) = c / (1 + 2^(-g log2 e)) on the hardware exponential and reciprocal (v_exp_f32 / v_rcp_f32, ~1 ulp each: the libm forms
// are ~50 VALU instructions per value, 21 k values per tile -- a third of the first version's tile time), the multiplications and the addition as PACKED fp32 operations (two values
// per v_pk_mul_f32 / v_pk_add_f32: the block is bound by its vector instructions); the result is rounded to fp16.  Same operations per value as the scalar form: same bits.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f16x8 gate8(const f32x4& c0, const f32x4& c1, const f32x4& g0, const f32x4& g1) {
    f16x8 v;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4& c = h ? c1 : c0;
            const f32x4& g = h ? g1 : g0;
            const f32x2 x = f32x2{g[2 * q], g[2 * q + 1]} * f32x2{-1.44269504088896340736f, -1.44269504088896340736f};
            const f32x2 d = f32x2{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])} + f32x2{1.0f, 1.0f};
            const f32x2 r = f32x2{c[2 * q], c[2 * q + 1]} * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
            v[4 * h + 2 * q] = (f16)r[0]; v[4 * h + 2 * q + 1] = (f16)r[1];
        }
    return v;
}

// The block's weights global -> LDS by LDS-DMA: every wave issues its 1-KB pieces back to back (one round trip for the whole blob instead of a load -> store loop's several;
// the blob is L2-resident after the first workgroups) -- the caller waits (s_waitcnt vmcnt(0)) and meets the barrier.
template <int NWAVES>
__device__ __forceinline__ void weights_to_lds(char* smem, const char* w, int bytes, int wave, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, bytes, 0x00020000);
    for (int q = wave; q * 1024 < bytes; q += NWAVES)
        if (q * 1024 + lane * 16 < bytes)                    // (the last piece: only the lanes inside the blob -- what follows it in LDS belongs to somebody else)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(smem + q * 1024), 16, q * 1024 + lane * 16, 0, 0, 0);
#endif
}

// Every LDS operand read is UNCONDITIONAL: a lane whose fragment is structurally zero (k-octet 3 of a 20-channel operand, a panel row beyond the real
// ones) reads the 16 zero bytes at OFF_ZERO instead (a broadcast).  A predicated read is a branch around a ds_read + s_waitcnt per fragment -- the first
// version of this kernel waited out the LDS latency once per tap (21 us per tile against 4 us of MFMA work).
__device__ __forceinline__ f16x8 lds16(const char* smem, int off, bool real) { return *(const f16x8*)(smem + (real ? off : OFF_ZERO)); }
// A PIXEL fragment whose k-octet is structural padding (octet 3 of a 24-channel A / B / Y pixel, octets 1 .. 3 of x's second k-step) needs no select (round 6, VERDICT r5
// item 3b: 250 compares + 275 selects per 260 MFMAs): the weight fragment it meets holds zeros there (lds16 above), so the lane may read whatever FINITE fp16 data follows its
// pixel -- the next pixel's first octet(s), the zero-filled slack behind the X tile, the first bytes of the region behind A, or the zeroed tail behind B (see the kernel's start).
__device__ __forceinline__ f16x8 px16(const char* smem, int off) { return *(const f16x8*)(smem + off); }

// A 20 -> 20 channel 3x3 conv over this wave's RW output rows x 32 pixels (PW = 2 RW pixel tiles: tile u = row * 2 + segment) of an LDS image (octet planes of
// `plane` bytes, 36-pixel rows, the output tile at halo offset (2, 2)): pb = the lane's pixel-fragment base (the plane of its octet + the first tap's pixel of the wave's
// first row, segment 0), w0 / w1 = its fragment offsets inside a tap block (pan_scpa_layout.h), woff = the conv's panels.  Column by column (dx outer, dy inner: the tap
// order of conv3x3_pc's fragment walk): the column's six weight fragments and the RW + 2 input rows' pixel fragments are read up front -- every pixel fragment once, used
// by up to three output rows -- then the MFMAs.  Every address is a lane constant + an immediate.
template <int RW>
__device__ __forceinline__ void conv33_rows(const char* smem, int pb, int w0, int w1, int woff, f32x4 (&acc)[2 * RW][2]) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        f16x8 w[3][2], b[RW + 2][2];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int t = 0; t < 2; ++t) w[dy][t] = px16(smem, (t ? w1 : w0) + woff + (dy * 3 + dx) * K_TAP);
#pragma unroll
        for (int rr = 0; rr < RW + 2; ++rr)
#pragma unroll
            for (int seg = 0; seg < 2; ++seg) b[rr][seg] = px16(smem, pb + (rr * HC + seg * 16 + dx) * 16);
#pragma unroll
        for (int rr = 0; rr < RW + 2; ++rr)
#pragma unroll
            for (int seg = 0; seg < 2; ++seg)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int lo = rr - dy;
                    if (lo >= 0 && lo < RW) {
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            acc[lo * 2 + seg][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[dy][t], b[rr][seg], acc[lo * 2 + seg][t], 0, 0, 0);
                    }
                }
        __builtin_amdgcn_sched_barrier(0);          // one column's fragments in registers at a time (hoisting all three spills; the SIMD's other wave fills the gap)
    }
}

template <int TH>
__global__ __launch_bounds__(512, 1) void pan_scpa_fused(const ScpaKP p) {
    constexpr int HR = TH + 4, NPX = HR * HC;
    static_assert(NPX % 16 == 0 && TH % 8 == 0, "whole 16-pixel MFMA tiles, whole rows per wave");
    constexpr int NP1 = NPX / 16;                           // conv1's pixel tiles: the whole halo region
    constexpr int YW = TW + 2, NY = (TH + 2) * YW;          // Y: rows 1 .. TH + 2, columns 1 .. TW + 2 of the halo tile -- what k4 reads (round 5: every column before, 41 tiles
    constexpr int NMID = (NY + 15) / 16;                    //    of 16 pixels for TH = 16; 39 now: five per wave instead of six)
    constexpr int PW = TH / 4, RW = PW / 2;                 // output pixel tiles (16 px) / rows per wave: TH rows x 2 segments over 8 waves
    constexpr int XQ = (NPX * 5 + 63) / 64;                 // 1-KiB LDS-DMA pieces of an X tile
    constexpr int KQ = (XQ + 7) / 8;
    constexpr int PL = NPX * 16;                            // an octet plane of the A / B / Y images (pan_scpa_layout.h)
    constexpr int XOFF = (W_BYTES + 255) / 256 * 256, AOFF = XOFF + XQ * 1024, BOFF = AOFF + 3 * PL;
    static_assert(PL % 256 == 0 && BOFF + 3 * PL + 1024 <= 160 * 1024, "LDS");
    constexpr int OOB = (int)0x80000000;
    constexpr int TB = 3;                                   // pixel tiles of P1 / P2b in flight per wave (their fragment reads are issued together)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const X = smem + XOFF;
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
    const long gbytes = p.G * 2;

    // per-lane source offsets of this wave's X pieces relative to the tile's first halo pixel (group 0): slot i = piece * 64 + lane -> pixel i / 5, 16-byte
    // part i % 5 (0..3: channels 0..31 of group 0; 4: channels 32..39 = the first 16 bytes of group 1)
    // (compact input, p.in_c8: part 4 lies at gbytes + pixel * 16 from the slab's origin = gbytes + (r W + c) * 16 - 48 * (the tile's first halo pixel index) from the tile's
    //  group-0 address: the per-lane part here, the per-tile part in fetch)
    int loff[KQ];
    [[maybe_unused]] unsigned part4 = 0;                    // bit k: piece k of this lane is part 4
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        const int i = (wave + 8 * k) * 64 + lane, P = i / 5, s = i - 5 * P, r = P / HC, c = P - r * HC;
        const bool ok = wave + 8 * k < XQ && P < NPX;
        loff[k] = ok ? (s < 4 ? (r * p.W + c) * 64 + s * 16 : (r * p.W + c) * (p.in_c8 ? 16 : 64) + (int)gbytes) : OOB;
        part4 |= (unsigned)(ok && s == 4) << k;
    }
    auto decode = [&](int j, int& n, int& ty0, int& tx0) __attribute__((always_inline)) {
        int t = run_start + j;
        n = t / per_img; t -= n * per_img;
        const int ty = t / p.tiles_x;
        ty0 = ty * TH; tx0 = (t - ty * p.tiles_x) * TW;
    };
    auto fetch = [&](int j) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        int n, ty0, tx0;
        decode(j, n, ty0, tx0);
        const char* src = (const char*)(p.in + (long)n * p.H * p.W * 32) + ((long)(ty0 - 2) * p.W + (tx0 - 2)) * 64;
        const bool edge = ty0 < 2 || ty0 + TH + 2 > p.H || tx0 < 2 || tx0 + TW + 2 > p.W;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
        const int c8_shift = p.in_c8 ? 48 * (int)((long)n * p.H * p.W + (long)(ty0 - 2) * p.W + (tx0 - 2)) : 0;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const int q = wave + 8 * k;
            int vo = loff[k];
            if ((part4 >> k) & 1) vo -= c8_shift;
            if (edge) {
                const int i = q * 64 + lane, P = i / 5, r = P / HC, c = P - r * HC;
                const int y = ty0 - 2 + r, x = tx0 - 2 + c;
                if (y < 0 || y >= p.H || x < 0 || x >= p.W) vo = OOB;
            }
            if (q < XQ) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(X + q * 1024), 16, vo, 0, 0, 0);
        }
#else
        (void)j;
#endif
    };

    fetch(j0);
    weights_to_lds<8>(smem, p.w, W_BYTES, wave, lane);
    // (finite data wherever a padding k-octet may be read: the 1 KB behind B -- px16 -- and, for the first tile's P1, the head of A behind the X tile's own zero-filled slack)
    if (tid < 64) *(f16x8*)(smem + BOFF + 3 * PL + tid * 16) = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (tid < 4) *(f16x8*)(smem + AOFF + tid * 16) = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const int row0 = RW * wave;                             // this wave's first output row of the tile (P2a / P3)
    // lane constants of the fragment reads (pan_scpa_layout.h): the octet plane a lane reads (k-octet 3: octet 2 again), its offsets inside a tap block, its pixel base
    const int lo_w = lg_w < 3 ? lg_w : 2, w0_w = kfrag_t0(li_w, lg_w), w1_w = kfrag_t1(li_w, lg_w);
    const int pconv_w = lo_w * PL + ((row0 + 1) * HC + 1 + li_w) * 16;
    ST_DECL
    for (int j = j0; j < run_len; j += slots) {
        ST_BEGIN
        int n, ty0, tx0;
        decode(j, n, ty0, tx0);
        // (opaque copies of the lane coordinates, renewed per tile: every LDS address below is then recomputed where it is used -- an add on a shared
        //  per-lane base and a select -- instead of being hoisted out of the tile loop as ~150 loop-invariant address registers, which spilled)
        int li = li_w, lg = lg_w;
        asm volatile("" : "+v"(li), "+v"(lg));
        // ---------------- P1: A | B = lrelu(conv1_a | conv1_b (x)) over the halo tile ----------------
#ifdef INNFER_ABLATE
        if (!(p.abl & 1))
#endif
        {
            f16x8 wa[2][2], wb[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ba = OFF_C1A + (t ? C1_T1 : 0) + li * C1_ROW, bb = OFF_C1B + (t ? C1_T1 : 0) + li * C1_ROW;
                wa[t][0] = lds16(smem, ba + lg * 16, li < r2(t)); wa[t][1] = lds16(smem, ba + 64, li < r2(t) && lg == 0);
                wb[t][0] = lds16(smem, bb + lg * 16, li < r2(t)); wb[t][1] = lds16(smem, bb + 64, li < r2(t) && lg == 0);
            }
            for (int i0 = wave; i0 < NP1; i0 += 8 * TB) {
                f16x8 b0[TB], b1[TB];
#pragma unroll
                for (int k = 0; k < TB; ++k) {
                    const int i = i0 + 8 * k < NP1 ? i0 + 8 * k : i0, P = 16 * i + li;        // (a tile past the end repeats the first one: same values stored twice)
                    b0[k] = *(const f16x8*)(X + P * 80 + lg * 16);
                    b1[k] = px16(smem, XOFF + P * 80 + 64 + lg * 16);
                }
#pragma unroll
                for (int k = 0; k < TB; ++k) {
                    const int i = i0 + 8 * k < NP1 ? i0 + 8 * k : i0, P = 16 * i + li;
                    f32x4 ca[2], cb[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        ca[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][0], b0[k], z4, 0, 0, 0);
                        ca[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][1], b1[k], ca[t], 0, 0, 0);
                        cb[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[t][0], b0[k], z4, 0, 0, 0);
                        cb[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[t][1], b1[k], cb[t], 0, 0, 0);
                    }
                    if (lg < 3) {
                        *(f16x8*)(smem + AOFF + lg * PL + P * 16) = lrelu8(ca[0], ca[1]);
                        *(f16x8*)(smem + BOFF + lg * PL + P * 16) = lrelu8(cb[0], cb[1]);
                    }
                }
            }
        }
        __syncthreads();
        ST_MARK(0)
        // ---------------- P2a: a' = lrelu(k1(A)) on the tile's own pixels (registers) ----------------
        f16x8 ap[PW], res[PW][2];
        {
            f32x4 acc[PW][2];
#pragma unroll
            for (int u = 0; u < PW; ++u) acc[u][0] = acc[u][1] = z4;
#ifdef INNFER_ABLATE
            if (!(p.abl & 2))
#endif
            conv33_rows<RW>(smem, AOFF + pconv_w, w0_w, w1_w, OFF_K1, acc);
#pragma unroll
            for (int u = 0; u < PW; ++u) ap[u] = lrelu8(acc[u][0], acc[u][1]);
            // the residual x of this wave's output pixels, in conv3's result layout (a lane: channels 16 lg .. 16 lg + 15 of pixel li), before X is re-used
#pragma unroll
            for (int u = 0; u < PW; ++u) {
                const int P = (row0 + (u >> 1) + 2) * HC + 2 + (u & 1) * 16 + li;
                res[u][0] = lds16(smem, XOFF + P * 80 + (lg < 2 ? 2 * lg : 4) * 16, lg < 3);
                res[u][1] = lds16(smem, XOFF + P * 80 + (2 * lg + 1) * 16, lg < 2);
            }
        }
        __syncthreads();                                            // every wave has read A (Y may take its place) and X (the next tile's may)
        ST_MARK(1)
#ifdef INNFER_ABLATE
        if (!(p.abl & 16))
#endif
        if (j + slots < run_len) fetch(j + slots);                 // the next tile's X lands while P2b / P3 run
        // ---------------- P2b: Y = k3(B) * sigmoid(k2(B) + bias) on rows 1 .. TH + 2, zero outside the image ----------------
#ifdef INNFER_ABLATE
        if (!(p.abl & 4))
#endif
        {
            const f32x4 bk0 = *(const f32x4*)(smem + OFF_B2 + (8 * lg) * 4), bk1 = *(const f32x4*)(smem + OFF_B2 + (8 * lg + 4) * 4);
            const bool edge_t = ty0 < 2 || ty0 + TH + 2 > p.H || tx0 < 2 || tx0 + TW + 2 > p.W;
            // pixel tile i of the flattened (TH + 2) x (TW + 2) region: pixel Q = 16 i + lane -> halo-tile pixel P = (Q / YW + 1) HC + Q % YW + 1
            auto halo_px = [&](int i) __attribute__((always_inline)) {
                const int Q = min(16 * i + li, NY - 1), r = Q / YW;
                return (r + 1) * HC + (Q - r * YW) + 1;
            };
            auto pass = [&](auto nkc, int i0) __attribute__((always_inline)) {      // NK pixel tiles i0, i0 + 8, .. of this wave, their fragment reads issued together
                constexpr int NK = decltype(nkc)::value;
                f32x4 c[NK][2], g[NK][2];
                int Pk[NK];
#pragma unroll
                for (int k = 0; k < NK; ++k) { c[k][0] = c[k][1] = z4; g[k][0] = bk0; g[k][1] = bk1; Pk[k] = halo_px(i0 + 8 * k < NMID ? i0 + 8 * k : i0); }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    f16x8 w[3][2], b[NK][3];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            w[dy][t] = px16(smem, (t ? w1_w : w0_w) + OFF_K3 + (dy * 3 + dx) * K_TAP);
#pragma unroll
                    for (int k = 0; k < NK; ++k)
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) b[k][dy] = px16(smem, BOFF + lo_w * PL + (Pk[k] - HC - 1) * 16 + (dy * HC + dx) * 16);
#pragma unroll
                    for (int k = 0; k < NK; ++k)
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int t = 0; t < 2; ++t)
                                c[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[dy][t], b[k][dy], c[k][t], 0, 0, 0);
                    if (dx == 1) {                                  // k2: the 1x1 conv of the gate reads the centre pixel's fragment
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const f16x8 w2 = px16(smem, (t ? w1_w : w0_w) + OFF_K2);
#pragma unroll
                            for (int k = 0; k < NK; ++k) g[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2, b[k][1], g[k][t], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int i = i0 + 8 * k < NMID ? i0 + 8 * k : i0, P = Pk[k];
                    f16x8 v = gate8(c[k][0], c[k][1], g[k][0], g[k][1]);
                    if (edge_t) {          // (a tile on the frame's border only -- wave-uniform: Y is zero outside the image = k4's zero padding; interior tiles pay no compare / select per value)
                        const int r = P / HC, cc = P - r * HC, y = ty0 - 2 + r, x = tx0 - 2 + cc;
                        if (!(y >= 0 && y < p.H && x >= 0 && x < p.W)) v = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    }
                    if (lg < 3 && 16 * i + li < NY) *(f16x8*)(smem + AOFF + lg * PL + P * 16) = v;
                }
            };
            // wave w takes tiles w, w + 8, ..: TB at a time; the last pass holds only the tiles that are left (two of 39 for TH = 16: six tile slots per wave were 41 tiles' worth)
            constexpr int NK0 = (NMID + 7) / 8 < TB ? (NMID + 7) / 8 : TB, NK1 = NMID > 8 * TB ? ((NMID - 8 * TB + 7) / 8 < TB ? (NMID - 8 * TB + 7) / 8 : TB) : 0;
            static_assert(NMID <= 16 * TB, "two passes cover the region");
            pass(std::integral_constant<int, NK0>{}, wave);
            if constexpr (NK1 > 0) pass(std::integral_constant<int, NK1>{}, wave + 8 * TB);
        }
        __syncthreads();
        ST_MARK(2)
        // ---------------- P3: b' = lrelu(k4(Y)); out = conv3(a' | b') + x ----------------
        {
            f32x4 acc[PW][2];
#pragma unroll
            for (int u = 0; u < PW; ++u) acc[u][0] = acc[u][1] = z4;
#ifdef INNFER_ABLATE
            if (!(p.abl & 8))
#endif
            conv33_rows<RW>(smem, AOFF + pconv_w, w0_w, w1_w, OFF_K4, acc);
            f16x8 w3[4][2];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    w3[t][ks] = lds16(smem, OFF_C3 + c3_t(t) + (li * 6 + ks * 3 + lg) * 16, li < r4(t) && lg < 3);
#pragma unroll
            for (int u = 0; u < PW; ++u) {
                const int row = row0 + (u >> 1), col = (u & 1) * 16 + li;
                const f16x8 bp = lrelu8(acc[u][0], acc[u][1]);
                f32x4 d[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    d[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3[t][0], ap[u], z4, 0, 0, 0);
                    d[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3[t][1], bp, d[t], 0, 0, 0);
                }
                const int y = ty0 + row, x = tx0 + col;
                // the next tile's X pieces (issued before P2b) must have landed before the barrier below; waiting HERE, in front of this tile's stores,
                // keeps the stores out of the wait (a vmcnt(0) behind them adds their round trip to every tile)
                if (u == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef INNFER_ABLATE
                if (p.abl & 32) continue;
#endif
                if (y < p.H && x < p.W) {
                    f16x8 o0, o1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o0[e] = (f16)(d[0][e] + (float)res[u][0][e]); o0[4 + e] = (f16)(d[1][e] + (float)res[u][0][4 + e]);
                        o1[e] = (f16)(d[2][e] + (float)res[u][1][e]); o1[4 + e] = (f16)(d[3][e] + (float)res[u][1][4 + e]);
                    }
                    const long pix = ((long)n * p.H + y) * p.W + x;
                    if (p.out_c8 && lg >= 2) {                       // channels 32..39 compact; 40..63 (zeros) not stored
                        if (lg == 2) *(f16x8*)(p.out + p.G + pix * 8) = o0;
                    } else {
                        f16* o = p.out + (lg >> 1) * p.G + pix * 32 + (lg & 1) * 16;
                        *(f16x8*)o = o0;
                        *(f16x8*)(o + 8) = o1;
                    }
                }
            }
        }
        __syncthreads();                                            // Y is dead, the next X is in LDS (every wave waited for its own pieces above)
        ST_MARK(3)
    }
    ST_END
}

// ---- the same block with TWO workgroups per CU (round 6, VERDICT r5 item 3b: "two tiles in flight per workgroup, or 2 x 4-wave workgroups per CU") ----------------
// What made room: x need not be staged in LDS (csrc/pan_scpa_split.hip showed it) -- conv1_a / conv1_b are 1x1, so a wave loads the B operand of its pixel tiles straight
// from memory in fragment layout, the next tile's while this tile's P3 runs -- which leaves the weights (35 KB) and the A | B planes.  A workgroup of FOUR waves owns an
// 8 x 32 tile (12 x 36 halo pixels: planes 41.5 KB): 77.6 KB of LDS, two workgroups per CU, each SIMD holding one wave of either.  The two run their barrier-separated
// phases independently, so one's vector-heavy P1 / P2b overlaps the other's MFMA-heavy P2a / P3 instead of both waves of a SIMD doing the same thing at the same time.
// Same arithmetic in the same order per value as pan_scpa_fused (k-steps, taps dx-major): bit-identical outputs.
constexpr int D_TH = 8, D_NPX = (D_TH + 4) * HC, D_NP1 = D_NPX / 16, D_PL = D_NPX * 16;           // 12 x 36 halo pixels = 27 pixel tiles; an octet plane
constexpr int D_YW = TW + 2, D_NY = (D_TH + 2) * D_YW, D_NMID = (D_NY + 15) / 16;                    // Y: the 10 x 34 region k4 reads = 22 pixel tiles
constexpr int D_KT = (D_NP1 + 3) / 4;                                                               // P1 tile slots per wave (7)
constexpr int D_W = (W_BYTES + 255) / 256 * 256, D_A = D_W, D_B = D_A + 3 * D_PL, D_LDS = D_B + 3 * D_PL + 1024;
static_assert(D_PL % 256 == 0 && 2 * D_LDS <= 160 * 1024, "two workgroups per CU");

__global__ __launch_bounds__(256, 2) void pan_scpa_duo(const ScpaKP p) {
    constexpr int OOB = (int)0x80000000;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li_w = lane & 15, lg_w = lane >> 4;
    const int bid = blockIdx.x, xcd = bid & 7;
    const int run_q = p.total >> 3, run_r = p.total & 7;
    const int run_start = xcd < run_r ? xcd * (run_q + 1) : run_r * (run_q + 1) + (xcd - run_r) * run_q;
    const int run_len = run_q + (xcd < run_r ? 1 : 0);
    const int slots = ((int)gridDim.x + 7 - xcd) >> 3;
    const int j0 = bid >> 3;
    if (j0 >= run_len) return;
    const int per_img = p.tiles_x * p.tiles_y;
    const int gbytes = (int)(p.G * 2), c8s = p.in_c8 ? 16 : 64;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0x7fffffff, 0x00020000);

    auto decode = [&](int j, int& n, int& ty0, int& tx0) __attribute__((always_inline)) {
        int t = run_start + j;
        n = t / per_img; t -= n * per_img;
        const int ty = t / p.tiles_x;
        ty0 = ty * D_TH; tx0 = (t - ty * p.tiles_x) * TW;
    };
    // the B operand of this wave's P1 pixel tiles (halo pixels 16 (wave + 4 k) + li of tile j): channels 0..31 (octet lg) and 32..39 (every lane group the same octet:
    // only k-octet 0 of the second k-step meets non-zero weights); pixels outside the image read zero (the buffer's range check) = the zero padding of the block's convs
    f16x8 x0[D_KT], x1[D_KT];
    auto load_x = [&](int j) __attribute__((always_inline)) {
        int n, ty0, tx0;
        decode(j, n, ty0, tx0);
        const int base = (n * p.H + ty0 - 2) * p.W + tx0 - 2;
        const bool edge = ty0 < 2 || ty0 + D_TH + 2 > p.H || tx0 < 2 || tx0 + TW + 2 > p.W;
        int li = li_w, lg = lg_w;
        asm volatile("" : "+v"(li), "+v"(lg));
#pragma unroll
        for (int k = 0; k < D_KT; ++k) {
            const int P = 16 * (wave + 4 * k) + li, r = P / HC, c = P - r * HC;
            int o0 = (base + r * p.W + c) * 64 + lg * 16, o1 = gbytes + (base + r * p.W + c) * c8s;
            if (edge) {
                const int y = ty0 - 2 + r, x = tx0 - 2 + c;
                if (y < 0 || y >= p.H || x < 0 || x >= p.W) o0 = o1 = OOB;
            }
            if (wave + 4 * k >= D_NP1) o0 = o1 = OOB;        // (an empty slot: the range check answers without a memory access)
            x0[k] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, o0, 0, 0));
            x1[k] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, o1, 0, 0));
        }
    };

    load_x(j0);
    weights_to_lds<4>(smem, p.w, W_BYTES, wave, lane);
    // (finite data wherever a fragment read may land: P1 writes every pixel of all six planes before the first read, so only the 1 KB behind them -- where the reads of the
    //  last pixels' right-hand taps end -- needs initialising)
    if (tid < 64) *(f16x8*)(smem + D_B + 3 * D_PL + tid * 16) = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const int row0 = 2 * wave;                               // this wave's two output rows (P2a / P3), both segments: pixel tile u = row * 2 + segment
    const int lo_w = lg_w < 3 ? lg_w : 2, w0_w = kfrag_t0(li_w, lg_w), w1_w = kfrag_t1(li_w, lg_w);
    const int pconv_w = lo_w * D_PL + ((row0 + 1) * HC + 1 + li_w) * 16;
    ST_DECL
    for (int j = j0; j < run_len; j += slots) {
        ST_BEGIN
        int n, ty0, tx0;
        decode(j, n, ty0, tx0);
        int li = li_w, lg = lg_w;
        asm volatile("" : "+v"(li), "+v"(lg));
        // ---------------- P1: A | B = lrelu(conv1_a | conv1_b (x)) over the halo tile ----------------
        {
            f16x8 wa[2][2], wb[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ba = OFF_C1A + (t ? C1_T1 : 0) + li * C1_ROW, bb = OFF_C1B + (t ? C1_T1 : 0) + li * C1_ROW;
                wa[t][0] = lds16(smem, ba + lg * 16, li < r2(t)); wa[t][1] = lds16(smem, ba + 64, li < r2(t) && lg == 0);
                wb[t][0] = lds16(smem, bb + lg * 16, li < r2(t)); wb[t][1] = lds16(smem, bb + 64, li < r2(t) && lg == 0);
            }
#pragma unroll
            for (int k = 0; k < D_KT; ++k) {
                if (wave + 4 * k >= D_NP1) continue;
                const int P = 16 * (wave + 4 * k) + li;
                f32x4 ca[2], cb[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    ca[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][0], x0[k], z4, 0, 0, 0);
                    ca[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][1], x1[k], ca[t], 0, 0, 0);
                    cb[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[t][0], x0[k], z4, 0, 0, 0);
                    cb[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[t][1], x1[k], cb[t], 0, 0, 0);
                }
                if (lg < 3) {
                    *(f16x8*)(smem + D_A + lg * D_PL + P * 16) = lrelu8(ca[0], ca[1]);
                    *(f16x8*)(smem + D_B + lg * D_PL + P * 16) = lrelu8(cb[0], cb[1]);
                }
            }
        }
        __syncthreads();
        ST_MARK(0)
        // ---------------- P2a: a' = lrelu(k1(A)) on the wave's own pixels (registers) ----------------
        f16x8 ap[4];
        {
            f32x4 acc[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u][0] = acc[u][1] = z4;
            conv33_rows<2>(smem, D_A + pconv_w, w0_w, w1_w, OFF_K1, acc);
#pragma unroll
            for (int u = 0; u < 4; ++u) ap[u] = lrelu8(acc[u][0], acc[u][1]);
        }
        __syncthreads();                                            // every wave has read A: Y may take its place
        ST_MARK(1)
        // ---------------- P2b: Y = k3(B) * sigmoid(k2(B) + bias) on rows 1 .. TH + 2, zero outside the image ----------------
        {
            const f32x4 bk0 = *(const f32x4*)(smem + OFF_B2 + (8 * lg) * 4), bk1 = *(const f32x4*)(smem + OFF_B2 + (8 * lg + 4) * 4);
            const bool edge_t = ty0 < 2 || ty0 + D_TH + 2 > p.H || tx0 < 2 || tx0 + TW + 2 > p.W;
            auto halo_px = [&](int i) __attribute__((always_inline)) {
                const int Q = min(16 * i + li, D_NY - 1), r = Q / D_YW;
                return (r + 1) * HC + (Q - r * D_YW) + 1;
            };
            auto pass = [&](int i0) __attribute__((always_inline)) {               // three pixel tiles i0, i0 + 4, i0 + 8 of this wave, their fragment reads issued together
                constexpr int NK = 3;
                f32x4 c[NK][2], g[NK][2];
                int Pk[NK];
#pragma unroll
                for (int k = 0; k < NK; ++k) { c[k][0] = c[k][1] = z4; g[k][0] = bk0; g[k][1] = bk1; Pk[k] = halo_px(i0 + 4 * k < D_NMID ? i0 + 4 * k : i0); }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    f16x8 w[3][2], b[NK][3];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int t = 0; t < 2; ++t) w[dy][t] = px16(smem, (t ? w1_w : w0_w) + OFF_K3 + (dy * 3 + dx) * K_TAP);
#pragma unroll
                    for (int k = 0; k < NK; ++k)
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) b[k][dy] = px16(smem, D_B + lo_w * D_PL + (Pk[k] - HC - 1) * 16 + (dy * HC + dx) * 16);
#pragma unroll
                    for (int k = 0; k < NK; ++k)
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int t = 0; t < 2; ++t)
                                c[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[dy][t], b[k][dy], c[k][t], 0, 0, 0);
                    if (dx == 1) {                                  // k2: the 1x1 conv of the gate reads the centre pixel's fragment
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const f16x8 w2 = px16(smem, (t ? w1_w : w0_w) + OFF_K2);
#pragma unroll
                            for (int k = 0; k < NK; ++k) g[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2, b[k][1], g[k][t], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int i = i0 + 4 * k < D_NMID ? i0 + 4 * k : i0, P = Pk[k];
                    f16x8 v = gate8(c[k][0], c[k][1], g[k][0], g[k][1]);
                    if (edge_t) {          // (a tile on the frame's border only: Y is zero outside the image = k4's zero padding)
                        const int r = P / HC, cc = P - r * HC, y = ty0 - 2 + r, x = tx0 - 2 + cc;
                        if (!(y >= 0 && y < p.H && x >= 0 && x < p.W)) v = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    }
                    if (lg < 3 && 16 * i + li < D_NY) *(f16x8*)(smem + D_A + lg * D_PL + P * 16) = v;
                }
            };
            static_assert(D_NMID <= 24, "two passes of three tile slots per wave cover the region");
            pass(wave);
            pass(wave + 12);
        }
        __syncthreads();
        ST_MARK(2)
        // ---------------- P3: b' = lrelu(k4(Y)); out = conv3(a' | b') + x ----------------
        {
            // the residual x of the wave's pixels in conv3's result layout (a lane: channels 16 lg .. 16 lg + 15 of pixel li), re-read from memory (L2)
            f16x8 res[4][2];
            long opix[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int y = ty0 + row0 + (u >> 1), x = tx0 + (u & 1) * 16 + li;
                const bool ok = y < p.H && x < p.W;
                const int pix = (n * p.H + y) * p.W + x;
                opix[u] = ok ? pix : -1;
                const int o0 = !ok || lg == 3 ? OOB : (lg < 2 ? pix * 64 + lg * 32 : gbytes + pix * c8s), o1 = ok && lg < 2 ? o0 + 16 : OOB;
                res[u][0] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, o0, 0, 0));
                res[u][1] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, o1, 0, 0));
            }
            load_x(j + slots < run_len ? j + slots : j);           // the next tile's x, in flight while this tile finishes (unconditional: a conditional load keeps the OLD x alive through P2)
            f32x4 acc[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u][0] = acc[u][1] = z4;
            conv33_rows<2>(smem, D_A + pconv_w, w0_w, w1_w, OFF_K4, acc);
            f16x8 w3[4][2];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    w3[t][ks] = lds16(smem, OFF_C3 + c3_t(t) + (li * 6 + ks * 3 + lg) * 16, li < r4(t) && lg < 3);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f16x8 bp = lrelu8(acc[u][0], acc[u][1]);
                f32x4 d[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    d[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3[t][0], ap[u], z4, 0, 0, 0);
                    d[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3[t][1], bp, d[t], 0, 0, 0);
                }
                if (opix[u] >= 0) {
                    f16x8 o0, o1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o0[e] = (f16)(d[0][e] + (float)res[u][0][e]); o0[4 + e] = (f16)(d[1][e] + (float)res[u][0][4 + e]);
                        o1[e] = (f16)(d[2][e] + (float)res[u][1][e]); o1[4 + e] = (f16)(d[3][e] + (float)res[u][1][4 + e]);
                    }
                    const long pix = opix[u];
                    if (p.out_c8 && lg >= 2) {                       // channels 32..39 compact; 40..63 (zeros) not stored
                        if (lg == 2) *(f16x8*)(p.out + p.G + pix * 8) = o0;
                    } else {
                        f16* o = p.out + (lg >> 1) * p.G + pix * 32 + (lg & 1) * 16;
                        *(f16x8*)o = o0;
                        *(f16x8*)(o + 8) = o1;
                    }
                }
            }
        }
        __syncthreads();                                            // Y is dead
        ST_MARK(3)
    }
    ST_END
}

}  // namespace

size_t pan_scpa_blob_bytes() { return W_BYTES; }

// conv1_a / conv1_b [20][40], k1 / k3 / k4 [20][20][3][3], k2 [20][20] + bias [20], conv3 [40][40] (torch layouts, fp32) -> the kernel's blob
void pan_scpa_pack(const float* c1a, const float* c1b, const float* k1, const float* k2, const float* k2b, const float* k3, const float* k4, const float* c3, void* blob) {
    char* w = (char*)blob;
    for (int i = 0; i < W_BYTES; ++i) w[i] = 0;
    auto put = [&](int off, int e, float v) { ((f16*)(w + off))[e] = (f16)v; };
    for (int part = 0; part < 2; ++part) {
        const float* src = part ? c1b : c1a;
        for (int t = 0; t < 2; ++t)
            for (int rho = 0; rho < r2(t); ++rho) {
                const int co = 8 * (rho >> 2) + 4 * t + (rho & 3);
                for (int oct = 0; oct < 5; ++oct)
                    for (int e = 0; e < 8; ++e)
                        put((part ? OFF_C1B : OFF_C1A) + (t ? C1_T1 : 0) + rho * C1_ROW + oct * 16, e, src[co * 40 + oct * 8 + e]);
            }
    }
    // a tap block of a 20 -> 20 conv: tile 0's 12 rows at slots 3 sigma0(row) + octet, tile 1's 8 rows at 36 + 3 sigma1(row) + octet, two zero slots (pan_scpa_layout.h)
    auto tap_block = [&](int base, const float* wsrc, int ntap, int tap) {
        for (int t = 0; t < 2; ++t)
            for (int rho = 0; rho < r2(t); ++rho) {
                const int co = 8 * (rho >> 2) + 4 * t + (rho & 3);
                const int slot = t ? 36 + 3 * SIG1[rho] : 3 * SIG0[rho];
                for (int oct = 0; oct < 3; ++oct)
                    for (int e = 0; e < 8; ++e) {
                        const int ci = oct * 8 + e;
                        put(base + (slot + oct) * 16, e, ci < 20 ? wsrc[(co * 20 + ci) * ntap + tap] : 0.f);
                    }
            }
    };
    const float* ks[3] = {k1, k3, k4};
    const int offs[3] = {OFF_K1, OFF_K3, OFF_K4};
    for (int c = 0; c < 3; ++c)
        for (int tap = 0; tap < 9; ++tap) tap_block(offs[c] + tap * K_TAP, ks[c], 9, tap);
    tap_block(OFF_K2, k2, 1, 0);
    for (int t = 0; t < 4; ++t)
        for (int rho = 0; rho < r4(t); ++rho) {
            const int co = 16 * (rho >> 2) + 4 * t + (rho & 3);
            for (int oct = 0; oct < 6; ++oct)
                for (int e = 0; e < 8; ++e) {
                    const int ci = (oct % 3) * 8 + e;                           // a' (octets 0..2) | b' (octets 3..5): cat[a, b] = input channels 0..19 | 20..39
                    put(OFF_C3 + c3_t(t) + rho * C3_ROW + oct * 16, e, ci < 20 ? c3[co * 40 + (oct / 3) * 20 + ci] : 0.f);
                }
        }
    for (int c = 0; c < 20; ++c) ((float*)(w + OFF_B2))[c] = k2b[c];
}

static int scpa_num_cus() {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    static std::atomic<int> cached[64] = {};              // (relaxed: every thread that misses stores the same value)
    v = cached[dev & 63].load(std::memory_order_relaxed);
    if (!v) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cached[dev & 63].store(v, std::memory_order_relaxed);
    }
    return v;
}

int pan_scpa_launch(const f16* in, f16* out, long G, const void* d_blob, int N, int H, int W, hipStream_t s, int in_c8, int out_c8, int duo) {
    const int num_cus = scpa_num_cus();
    constexpr int TH = 16;
    constexpr int LDS = 160 * 1024;
    static std::atomic<unsigned long long> attr_done{0};    // function attributes belong to the device's copy of the code object; concurrent first calls may both set it (idempotent)
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(attr_done.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute((const void*)pan_scpa_fused<TH>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
                (void)hipGetLastError();
                return set_error(INNFER_ERR_UNSUPPORTED, "pan_scpa: the fused SCPA block needs 160 KB of LDS per workgroup (gfx950); innfer_pan_set_fused_scpa(pan, 0) selects the five-launch schedule");
            }
            attr_done.fetch_or(bit, std::memory_order_release);
        }
    }
    if ((long)N * H * W * 64 + 2 * G >= 0x7fffffffL) return set_error(INNFER_ERR_UNSUPPORTED, "pan_scpa: slab too large for 32-bit buffer offsets");
    ScpaKP k{};
    k.in = in; k.out = out; k.G = G; k.w = (const char*)d_blob; k.N = N; k.H = H; k.W = W;
    k.in_c8 = in_c8; k.out_c8 = out_c8;
#ifdef INNFER_STAMPS
    k.stamps = g_scpa_stamps;
#endif
    k.tiles_x = (W + TW - 1) / TW; k.tiles_y = (H + TH - 1) / TH;
    const long total = (long)N * k.tiles_x * k.tiles_y;
    if (total > 0x7fffffffL) return set_error(INNFER_ERR_INVALID, "pan_scpa: grid too large");
    k.total = (int)total;
#ifdef INNFER_ABLATE
    k.abl = getenv("INNFER_SCPA_ABL") ? atoi(getenv("INNFER_SCPA_ABL")) : 0;
#endif
    // duo < 0: the default form -- two workgroups per CU.  It wins most where 16-row tiles would waste half of their last row of tiles or more (the command line's 200 x 200 chop
    // tiles are 12.5 tiles high: -10 % per launch) or leave CUs without a tile, and since its vector work shrank (docs/EXPERIMENTS.md 127-131) also at 540 x 960 (-3.6 %)
    if (duo < 0) duo = 1;
    if (duo) {           // two 4-wave workgroups per CU on 8 x 32 tiles (pan_scpa_duo)
        static std::atomic<unsigned long long> duo_done{0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(duo_done.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute((const void*)pan_scpa_duo, hipFuncAttributeMaxDynamicSharedMemorySize, D_LDS) != hipSuccess) {
                (void)hipGetLastError();
                return set_error(INNFER_ERR_UNSUPPORTED, "pan_scpa: the two-workgroup form needs 78 KB of LDS per workgroup");
            }
            duo_done.fetch_or(bit, std::memory_order_release);
        }
        k.tiles_y = (H + D_TH - 1) / D_TH;
        const long tot8 = (long)N * k.tiles_x * k.tiles_y;
        if (tot8 > 0x7fffffffL) return set_error(INNFER_ERR_INVALID, "pan_scpa: grid too large");
        k.total = (int)tot8;
        const int g2 = tot8 < 2L * num_cus ? (int)tot8 : 2 * num_cus;
        hipLaunchKernelGGL(pan_scpa_duo, dim3(g2), dim3(256), D_LDS, s, k);
        INNFER_HIP(hipGetLastError());
        return INNFER_OK;
    }
    const int grid = total < num_cus ? (int)total : num_cus;
    hipLaunchKernelGGL(pan_scpa_fused<TH>, dim3(grid), dim3(512), LDS, s, k);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

}  // namespace innfer
