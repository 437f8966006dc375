// Gather-GEMM on MFMA for the convolutions the halo-tile kernel (conv3x3.hip) cannot express -- strided, transposed, dilated, 7x7,
// reflection-padded (UNet, CycleGAN ResNet, PPON's dilated convs, WBC UNet, PAN's attention projections): out[pixel][co] = sum over taps and input
// channels of in[pixel displaced by the tap][ci] * W[tap][ci][co].  One 128-pixel x 64-channel tile per
// workgroup, operands staged through LDS with register-prefetched loads (the SR hot path uses
// conv3x3.hip).  Input: blocked-NHWC fp16 slab; output: fp32 [pixel][cout_pad].
#pragma once
#include "common.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace innfer {
namespace gg {

struct GP {
    const f16* in; long in_g; int nchunks; int N, Hin, Win;
    const f16* wpk;                       // [cot][tap][chunk][64 rows][64 B], the LDS image of the A operand
    float* out; int cout_pad;             // raw fp32 [N*Hfull*Wfull][raw_stride]; cout_pad/64 channel tiles are computed
    int raw_stride;                       // floats per pixel of the raw buffer (multiple of 4): channels beyond it are not stored; a stride
                                          // larger than cout_pad lets several GEMMs fill one row (out points at the first channel)
    int Ho, Wo, stride;                   // this launch's output grid; in = out*stride + d
    int ntaps; int dy[49], dx[49];        // up to 7x7 taps
    int ntaps_w; int wtap[49];            // taps of the weight panel, and the panel tap behind entry t of dy / dx: the launcher drops taps that lie in the
                                          // padding for EVERY output pixel of the grid (2x2 -> 1x1 under a 4x4 stride-2 kernel: 12 of 16), a zero product
    int reflect;                          // out-of-image taps read the mirrored pixel (nn.ReflectionPad2d) instead of zero
    int Hfull, Wfull, os, ooy, oox;       // out pixel = (oy*os + ooy, ox*os + oox)
    int up;                               // input is read through nearest-2x upsampling (Hin, Win = source size)
    int seg;                              // k-steps per split-K segment
    int ksplit; long split_elems;         // ksplit > 1: blockIdx.z computes segment z and writes out + z*split_elems
    int cout_store;                       // channels [0, cout_store) of the computed tile are written
    int ngroup; long g_wbytes; long g_outoff; int g_tapmul;   // ngroup > 1: blockIdx.z / ksplit selects one of several convs of the SAME
                                          // input: weights wpk + g*g_wbytes, taps * (1 + g*g_tapmul), output out + g*g_outoff
    int g_phase;                          // ngroup == 4 output phases of a stride-2 transposed conv: group g uses taps [g*ntaps, (g+1)*ntaps)
                                          // of dy/dx and writes output pixel (oy*os + (g>>1), ox*os + (g&1))
#ifdef INNFER_ABLATE
    int abl;                              // diagnostic build only (INNFER_GG_ABL): 1 = the pixel operand staged for a chunk's first tap only, 2 = the weight panels for the first steps only
#endif
};

// 128 pixels x 64 output channels per workgroup (4 waves x (32 px x 64 co) = 8 MFMAs per wave and 32-channel
// k-step).  Operands go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 12 one-KiB pieces per step)
// through a ring of GG_STAGES stages, issued GG_STAGES-1 steps ahead: a wave waits only for its own pieces
// of the current step (counted vmcnt), one raw barrier per step publishes them and frees the stage consumed
// last.  Out-of-image taps (the conv's zero padding / the transposed conv's missing taps) and pixels beyond
// M carry an offset beyond num_records: the buffer range check writes zeros to LDS.
// Layers with at most this many output pixels per image (and >= 32 k-steps) are split over K.
constexpr int SPLIT_MAX_PX = 256;
inline long split_max_px() {
    const long v = INNFER_KNOB("INNFER_SPLIT_PX", 64);
    return v < SPLIT_MAX_PX ? v : SPLIT_MAX_PX;
}

// Tile shapes: <BP, Q, STAGES, NCO> = 16*BP pixels per wave x 64*Q output channels per workgroup; 4*NCO waves, wave w multiplies pixel block w & 3
// with the Q / NCO panels of channel block w >> 2:
//   <2,1,4>    128 px x  64 co, 12 KiB per stage   small / narrow layers and every split-K launch
//   <4,2,3>    256 px x 128 co, 24 KiB per stage   big layers (twice the MFMAs per staged byte)
//   <4,4,3,2>  256 px x 256 co, 32 KiB per stage, 8 waves: the same MFMAs per wave as <4,2,3> with two thirds of the LDS-DMA bytes per MFMA -- the
//              launches sit on the CU's LDS-DMA fill rate (experiments 38 / 42), so bytes per MFMA is what counts (round 3)
template <int BP, int Q, int STAGES, int NCO = 1>
static __global__ __launch_bounds__(256 * NCO) void gemm_gather(const GP p) {
    constexpr int PXT = 64 * BP;                                      // pixels per workgroup
    constexpr int NW = 4 * NCO, QW = Q / NCO, BPW = BP / NCO;         // waves; panels a wave multiplies; pixel pieces a wave stages
    static_assert(Q % NCO == 0 && BP % NCO == 0, "panels and pixel pieces split evenly over the channel blocks");
    constexpr int B_BYTES = PXT * 64, A_BYTES = Q * 4096, STAGE_BYTES = B_BYTES + A_BYTES;
    constexpr int NPIECE = BPW + QW;                                  // LDS-DMA pieces per wave and step
    __shared__ __attribute__((aligned(16))) char lds[STAGES * STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int pw = wave & 3, cwv = wave >> 2;                         // this wave's pixel block and channel block
    const long M = (long)p.N * p.Ho * p.Wo;
    const long m0 = (long)blockIdx.x * PXT;
    const int cot = blockIdx.y * Q;                                   // first 64-channel panel of this workgroup

    // staging role: pixel rows (tid>>2) + 16*NW*h, 16-byte slot (tid&3); weight quarter (wave & 3) of panels cot + (wave >> 2) + NCO*j
    const int sslot = tid & 3;
    int spix[BPW], soy[BPW], sox[BPW], cso[BPW]; bool sok[BPW];
#pragma unroll
    for (int h = 0; h < BPW; ++h) {
        const int srow = (tid >> 2) + 16 * NW * h;
        const long sm = m0 + srow;
        sok[h] = sm < M;
        int n = 0;
        soy[h] = sox[h] = 0;
        if (sok[h]) {
            sox[h] = (int)(sm % p.Wo);
            soy[h] = (int)((sm / p.Wo) % p.Ho);
            n = (int)(sm / ((long)p.Wo * p.Ho));
        }
        spix[h] = n * p.Hin * p.Win;                                  // first pixel of the image (launch checks 32-bit range)
        cso[h] = (sslot ^ (((srow >> 2) & 1) << 1)) * 16;             // byte offset of the channel slot stored at LDS slot sslot
    }
    const long panel_bytes = (long)p.ntaps_w * p.nchunks * 4096;
    const int zg = p.ngroup > 1 ? (int)blockIdx.z / p.ksplit : 0, zs = (int)blockIdx.z - zg * p.ksplit;
    const int tapmul = 1 + zg * p.g_tapmul;
    const int tap0 = p.g_phase ? zg * p.ntaps : 0;
    const char* wtile = (const char*)p.wpk + zg * p.g_wbytes + (long)cot * panel_bytes;
    const int total_steps = p.ntaps * p.nchunks;
    // split-K: blockIdx.z takes the k-steps [z*seg, (z+1)*seg) and writes a partial result
    const int step0 = p.ksplit > 1 ? zs * p.seg : 0;
    const int nsteps = p.ksplit > 1 ? min(p.seg, total_steps - step0) : total_steps;

    auto issue = [&](int rel) {
        const int step = step0 + rel;
#if defined(__HIP_DEVICE_COMPILE__)
        const int t = step / p.nchunks, c = step - t * p.nchunks;
        char* st = lds + (rel % STAGES) * STAGE_BYTES;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (long)c * p.in_g), 0, 0x7fffffff, 0x00020000);
#ifdef INNFER_ABLATE
        // INNFER_GG_ABL 1 (wrong results by construction): the pixel operand is staged for the FIRST tap of a chunk only -- what a form that keeps the input tile in
        // LDS across the taps could gain at most (profiles/r4/gg_resident_bound.txt)
        if (!(p.abl & 1) || t == 0)
#endif
#pragma unroll
        for (int h = 0; h < BPW; ++h) {
            int iy = soy[h] * p.stride + p.dy[tap0 + t] * tapmul, ix = sox[h] * p.stride + p.dx[tap0 + t] * tapmul;
            if (p.reflect) {                                          // pad < size: one reflection is enough
                iy = iy < 0 ? -iy : (iy >= p.Hin ? 2 * p.Hin - 2 - iy : iy);
                ix = ix < 0 ? -ix : (ix >= p.Win ? 2 * p.Win - 2 - ix : ix);
            }
            bool ok = sok[h] && iy >= 0 && ix >= 0;
            if (p.up) { ok = ok && iy < 2 * p.Hin && ix < 2 * p.Win; iy >>= 1; ix >>= 1; }
            else ok = ok && iy < p.Hin && ix < p.Win;
            const int voff = ok ? (spix[h] + iy * p.Win + ix) * 64 + cso[h] : (int)0x80000000;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(st + (wave + NW * h) * 1024), 16, voff, 0, 0, 0);
        }
#ifdef INNFER_ABLATE
        if (!(p.abl & 2) || rel < STAGES)           // INNFER_GG_ABL 2: the weight panels staged for the first STAGES steps only (what the weight stream costs)
#endif
#pragma unroll
        for (int jj = 0; jj < QW; ++jj) {
            const int j = cwv + NCO * jj;
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(wtile + j * panel_bytes + ((long)p.wtap[tap0 + t] * p.nchunks + c) * 4096), 0, 4096, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(st + B_BYTES + j * 4096 + pw * 1024), 16, lane * 16, pw * 1024, 0, 0);
        }
#else
        (void)step; (void)rel; (void)wtile; (void)spix; (void)cso; (void)panel_bytes; (void)tapmul; (void)tap0;
#endif
    };

    f32x4 acc[BP][4 * QW];
#pragma unroll
    for (int h = 0; h < BP; ++h)
#pragma unroll
        for (int q = 0; q < 4 * QW; ++q) acc[h][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    int boff[BP];
#pragma unroll
    for (int h = 0; h < BP; ++h) {
        const int brow = pw * 16 * BP + 16 * h + li;
        boff[h] = brow * 64 + ((lg ^ (((brow >> 2) & 1) << 1)) << 4);
    }
    const int aoff = B_BYTES + cwv * QW * 4096 + li * 64 + ((lg ^ (((li >> 2) & 1) << 1)) << 4);      // (panels cwv*QW .. of the stage: see the note on the order below)

    for (int s0 = 0; s0 < STAGES - 1 && s0 < nsteps; ++s0) issue(s0);
    for (int step = 0; step < nsteps; ++step) {
        // wait for this wave's own pieces of `step`: up to STAGES-2 later steps stay in flight
        const int ahead = min(STAGES - 2, nsteps - 1 - step);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NPIECE) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPIECE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (step + STAGES - 1 < nsteps) issue(step + STAGES - 1);
        const char* b = lds + (step % STAGES) * STAGE_BYTES;
        f16x8 bf[BP];
#pragma unroll
        for (int h = 0; h < BP; ++h) bf[h] = *(const f16x8*)(b + boff[h]);
#pragma unroll
        for (int q = 0; q < 4 * QW; ++q) {
            const f16x8 a = *(const f16x8*)(b + q * 1024 + aoff);
#pragma unroll
            for (int h = 0; h < BP; ++h) acc[h][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bf[h], acc[h][q], 0, 0, 0);
        }
    }
    // D rows = out channels (16 (q & 3) + 4 lg + j of the 64-channel panel q >> 2), cols = pixels
#pragma unroll
    for (int h = 0; h < BP; ++h) {
        const long m = m0 + pw * 16 * BP + 16 * h + li;
        if (m < M) {
            const int ox = (int)(m % p.Wo);
            const int oy = (int)((m / p.Wo) % p.Ho);
            const long n = m / ((long)p.Wo * p.Ho);
            const int ooy = p.g_phase ? zg >> 1 : p.ooy, oox = p.g_phase ? zg & 1 : p.oox;
            const long opix = (n * p.Hfull + (long)oy * p.os + ooy) * p.Wfull + (long)ox * p.os + oox;
            float* op = p.out + (long)zs * p.split_elems + zg * p.g_outoff + opix * p.raw_stride;
#pragma unroll
            for (int q = 0; q < 4 * QW; ++q) {
                const int ch = (cot + cwv * QW + (q >> 2)) * 64 + 16 * (q & 3) + 4 * lg;      // (natural row order: see pack_panels)
                if (ch < p.cout_store) *(f32x4*)(op + ch) = acc[h][q];
            }
        }
    }
}


// ---- host: weight panels -----------------------------------------------------------------------
// row R = q*16 + rho of a 64-channel tile holds out channel cot*64 + R (the natural order: a lane group lg ends up with channels 16 q + 4 lg .. + 3 of MFMA tile q, so ONE
// store instruction -- 16 bytes per lane -- writes a pixel's 16 consecutive channels as 64 contiguous bytes.  Until round 4 the rows were permuted so that a lane
// held 16 consecutive channels over its four tiles: every instruction then wrote four 16-byte pieces 64 bytes apart per pixel, and the split-K partials -- 134 MB
// on the UNet's 8x8 -> 16x16 level -- paid for it: profiles/r4/gg_resident_bound.txt);
// LDS slot s of that row holds input channels chunk*32 + 8*(s ^ 2*bit2(R)) .. +7
template <typename W>
inline void pack_panels(std::vector<f16>& dst, int cout, int cin, int cin_pad, int ntaps, W weight_of) {
    const int cots = (cout + 63) / 64, nch = cin_pad / 32;
    dst.assign((size_t)cots * ntaps * nch * 64 * 32, (f16)0.f);
    size_t o = 0;
    for (int cot = 0; cot < cots; ++cot)
        for (int t = 0; t < ntaps; ++t)
            for (int c = 0; c < nch; ++c)
                for (int R = 0; R < 64; ++R) {
                    const int q = R >> 4, rho = R & 15;
                    const int co = cot * 64 + 16 * q + rho;
                    for (int s = 0; s < 4; ++s) {
                        const int cg = s ^ (((R >> 2) & 1) << 1);
                        for (int e = 0; e < 8; ++e, ++o) {
                            const int ci = c * 32 + cg * 8 + e;
                            if (co < cout && ci < cin) dst[o] = (f16)weight_of(co, ci, t);
                        }
                    }
                }
}


// raw[pixel][0:cout_pad] = sum over the ksplit partial results, in split order (deterministic)
static __global__ void splitk_reduce(const float* part, long split_elems, int ksplit, float* raw, int cout_pad,
                                     int N, int Ho, int Wo, int Hfull, int Wfull, int os, int ooy, int oox) {
    const int c4 = cout_pad >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long M = (long)N * Ho * Wo;
    if (i >= M * c4) return;
    const long m = i / c4;
    const int c = (int)(i - m * c4) * 4;
    const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho);
    const long n = m / ((long)Wo * Ho);
    const long o = ((n * Hfull + (long)oy * os + ooy) * Wfull + (long)ox * os + oox) * cout_pad + c;
    f32x4 a = *(const f32x4*)(part + o);
    for (int z = 1; z < ksplit; ++z) {
        const f32x4 b = *(const f32x4*)(part + z * split_elems + o);
        a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
    }
    *(f32x4*)(raw + o) = a;
}

// One launch.  dy/dx: ntaps displacements; stride: input step per output pixel (2 for the 4x4 s2 conv).
// scratch (optional): when the launch would not fill the chip (deep UNet layers: few pixels, a K loop of up
// to 256 steps) the K range is split over blockIdx.z into partial results that splitk_reduce adds up.
inline int launch(const f16* wpk, int cin_pad, int cout_pad, const f16* in, long in_g, int N, int Hin, int Win,
                  float* raw, int Ho, int Wo, int stride, int ntaps, const int* dy, const int* dx,
                  int Hfull, int Wfull, int os, int ooy, int oox, int up, hipStream_t s,
                  float* scratch = nullptr, size_t scratch_bytes = 0, int raw_stride = 0, int reflect = 0,
                  int ngroup = 1, long g_wbytes = 0, long g_outoff = 0, int g_tapmul = 0, int cout_store = 0, int g_phase = 0,
                  int* ksplit_out = nullptr) {      // ksplit_out: the caller reduces the partial results itself (scratch, *ksplit_out segments of
                                                     // N*Hfull*Wfull*raw_stride floats; 1 = not split, the result is in raw)
    GP g{};
    g.in = in; g.in_g = in_g; g.nchunks = cin_pad / 32; g.N = N; g.Hin = Hin; g.Win = Win;
    g.wpk = wpk; g.out = raw; g.cout_pad = cout_pad;
    g.raw_stride = raw_stride > 0 ? raw_stride : cout_pad;
    if (g.raw_stride <= 0 || (g.raw_stride & 3)) return set_error(INNFER_ERR_INVALID, "gather GEMM: bad raw stride %d", g.raw_stride);
    g.Ho = Ho; g.Wo = Wo; g.stride = stride; g.ntaps = ntaps;
    if (g_phase && (ngroup != 4 || ntaps * 4 > 49)) return set_error(INNFER_ERR_INVALID, "gather GEMM: phase groups need 4 groups of <= 12 taps");
    if (ntaps < 1 || ntaps > 49) return set_error(INNFER_ERR_INVALID, "gather GEMM: %d taps", ntaps);
    g.ntaps_w = ntaps;
    {   // Taps whose input pixel lies outside the image for every output pixel of the grid multiply zeros: drop them (the decision depends on the layer's
        // geometry only, never on the batch).  Phase groups keep equal tap counts (they share the k loop); otherwise nothing is dropped.
        const int ng = g_phase ? 4 : 1;
        int keep[49], cnt[4] = {0, 0, 0, 0};
        auto live = [&](int d, int no, int nin) {
            if (reflect) return true;
            for (int o = 0; o < no; ++o) { const int i = o * stride + d; if (i >= 0 && i < (up ? 2 * nin : nin)) return true; }
            return false;
        };
        for (int gph = 0; gph < ng; ++gph)
            for (int t = 0; t < ntaps; ++t) {
                const bool lv = (ngroup > 1 && !g_phase) || (live(dy[gph * ntaps + t], Ho, Hin) && live(dx[gph * ntaps + t], Wo, Win));   // (dilated groups scale the taps: kept)
                keep[gph * ntaps + t] = lv;
                cnt[gph] += lv;
            }
        bool same = cnt[0] > 0;
        for (int gph = 1; gph < ng; ++gph) same = same && cnt[gph] == cnt[0];
        int n_act = same ? cnt[0] : ntaps;
        for (int gph = 0; gph < ng; ++gph) {
            int o = 0;
            for (int t = 0; t < ntaps; ++t)
                if (!same || keep[gph * ntaps + t]) { g.dy[gph * n_act + o] = dy[gph * ntaps + t]; g.dx[gph * n_act + o] = dx[gph * ntaps + t]; g.wtap[gph * n_act + o] = t; ++o; }
        }
        ntaps = n_act;
        g.ntaps = ntaps;
    }
    g.Hfull = Hfull; g.Wfull = Wfull; g.os = os; g.ooy = ooy; g.oox = oox; g.up = up; g.reflect = reflect;
#ifdef INNFER_ABLATE
    g.abl = getenv("INNFER_GG_ABL") ? atoi(getenv("INNFER_GG_ABL")) : 0;
#endif
    const long M = (long)N * Ho * Wo;
    if (M <= 0) return INNFER_OK;
    // (launch timer: the GEMM and, where it is split over K inside this call, its reduction as one entry)
    GtScope gt(s, "gemm_gather (+ split-K reduce)", 2.0 * M * ngroup * (double)cout_pad * ntaps * cin_pad,
               (double)N * Hin * Win * cin_pad * 2.0 + (double)M * ngroup * cout_pad * 4.0 + (double)ngroup * ntaps * cin_pad * cout_pad * 2.0);
    if ((long)N * Hin * Win * 64 >= 0x7fffffffL) return set_error(INNFER_ERR_UNSUPPORTED, "gather GEMM: input of %d x %d x %d pixels exceeds the 2 GiB buffer window", N, Hin, Win);
    const int nsteps = ntaps * g.nchunks;
    const size_t full = (size_t)N * Hfull * Wfull * g.raw_stride * sizeof(float);
    // Split-K (deep UNet layers: a few pixels per image, a K loop of up to 256 steps): the decision depends on
    // the layer only, never on the batch size, so a batch stays bit-identical to the batch-1 forwards.
    // segments of >= 32 k-steps, at most 8: every segment costs a full-resolution fp32 partial result (written here, read back by the reduction), and
    // a ConvTranspose level of 128 k-steps split eight ways moved 2 x 268 MB of partials for a 69 GFLOP layer (UNet_256 x 64, 8x8 -> 16x16)
    // (grids of at most 4 pixels per image -- the 2x2 and 1x1 levels: their partial results are a few hundred KB, their launches a handful of workgroups
    //  walking a latency-bound k loop: segments of 8 k-steps, up to 32 of them)
    const bool tiny = (long)Ho * Wo <= 4;      // (measured: at 16 pixels per image the 32-way split loses to the 8-way one, 36 -> 45 us, and its reduction too)
    const int want = tiny ? std::min(32, std::max(1, nsteps / 8)) : nsteps >= 256 ? INNFER_KNOB("INNFER_GG_WANT256", 8) : nsteps >= 128 ? INNFER_KNOB("INNFER_GG_WANT128", 4) : nsteps >= 64 ? INNFER_KNOB("INNFER_GG_WANT64", 2) : 1;
    g.seg = (nsteps + want - 1) / want;
    const int segs = (nsteps + g.seg - 1) / g.seg;
    int ks = 1;
    if (scratch && (nsteps >= 32 || (tiny && segs > 1)) && (long)Ho * Wo <= split_max_px() && (size_t)segs * full <= scratch_bytes) ks = segs;
    if (ngroup > 1 && !g_phase) ks = 1;        // phase groups may be split over K: every segment buffer then holds the whole full-resolution grid
    g.ksplit = ks;
    g.cout_store = cout_store > 0 ? cout_store : (g.raw_stride < cout_pad ? g.raw_stride : cout_pad);
    g.ngroup = ngroup; g.g_wbytes = g_wbytes; g.g_outoff = g_outoff; g.g_tapmul = g_tapmul; g.g_phase = g_phase;
    g.split_elems = ks > 1 ? (long)(full / sizeof(float)) : 0;
    if (ks > 1) g.out = scratch;
    // (the tile shape never changes an element's accumulation order, so it may follow the batch size -- split launches included: 64 x 256^2
    //  1.76 -> 1.72 ms.  Folding the segments inside one workgroup instead of splitting -- same bits -- was measured too: the k loop of a
    //  workgroup is latency-bound, 108 -> 181 us on a 256-tile layer, and no gain beside the wide split tiles: not built)
    const bool big = !tiny && (ngroup == 1 || g_phase) && cout_pad % 128 == 0 &&
                     ((M + 255) / 256) * (cout_pad / 128) * ngroup * ks >= 256;
    // (256 x 256 tiles where they still give every CU a workgroup: two thirds of the LDS-DMA bytes per MFMA of the 256 x 128 form)
    const bool big2 = big && cout_pad % 256 == 0 && INNFER_KNOB("INNFER_GG_BIG2", 1) &&
                      ((M + 255) / 256) * (cout_pad / 256) * ngroup * ks >= 256;
    if (big2) {
        dim3 grid((unsigned)((M + 255) / 256), (unsigned)(cout_pad / 256), (unsigned)(ks * ngroup));
        hipLaunchKernelGGL((gemm_gather<4, 4, 3, 2>), grid, dim3(512), 0, s, g);
    } else if (big) {
        dim3 grid((unsigned)((M + 255) / 256), (unsigned)(cout_pad / 128), (unsigned)(ks * ngroup));
        hipLaunchKernelGGL((gemm_gather<4, 2, 3>), grid, dim3(256), 0, s, g);
    } else {
        dim3 grid((unsigned)((M + 127) / 128), (unsigned)(cout_pad / 64), (unsigned)(ks * ngroup));
        hipLaunchKernelGGL((gemm_gather<2, 1, 4>), grid, dim3(256), 0, s, g);
    }
    INNFER_HIP(hipGetLastError());
    if (ksplit_out) *ksplit_out = ks;
    if (ks > 1 && !ksplit_out) {
        // (phase groups: the four launches' worth of partials interleave into the full grid, reduced in one pass)
        const int rh = g_phase ? Hfull : Ho, rw = g_phase ? Wfull : Wo;
        const long nthr = (long)N * rh * rw * (g.raw_stride / 4);
        hipLaunchKernelGGL(splitk_reduce, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, (const float*)scratch, g.split_elems, ks,
                           raw, g.raw_stride, N, rh, rw, Hfull, Wfull, g_phase ? 1 : os, g_phase ? 0 : ooy, g_phase ? 0 : oox);
        INNFER_HIP(hipGetLastError());
    }
    return INNFER_OK;
}

}  // namespace gg
}  // namespace innfer
