// STATS / SGATE / RLDS pieces of conv3x3_pc: norm statistics out of the epilogue, the self gate (PAN), the residual from the live LDS stage.
// Part of csrc/conv3x3.hip (split out in round 5, VERDICT r4 item 7: no functional change -- the device assembly of the translation unit is identical);
// included there, inside namespace innfer { namespace { .. } }, after KP / the tile constants.  Not a stand-alone header.

// Partial statistics of a norm layer that follows the conv, out of the accumulators (fp32, bias included, before the fp16 rounding): every consumer
// wave reduces its RPW x 32 pixels per channel to (count, mean, M2 = sum of squared deviations from that mean) -- in-lane over its pixel tiles,
// a fixed xor butterfly over the 16 pixel lanes -- and writes them to
//   part[((slot * NCW + wave) * cn + channel) * 3],  slot = tile index over the batch (x 4 + phase behind the phase lattice).
// norm::combine_parts merges an image's partials in index order (Chan's update): deterministic, no atomics, and the pass that re-read the conv
// output for its statistics is gone.
template <int RPW, int NT, bool DCV, bool PAIR, int NCW>
__device__ __forceinline__ void epilogue_stats(const KP& p, const f32x4 (&acc)[NT][2 * RPW], const f32x4 (&bias)[NT], int ty0, int tx0, int wave, int li,
                                               int cbase, int tile) {
    static_assert(NT == 4, "sixteen channels per lane: one per pixel lane after the transposing reduction");
    constexpr int MT = 2 * RPW;
    int yw = ty0 + wave * RPW, x0 = tx0, ylim = p.y1, c0 = cbase, slot = tile, ph = 0;
    if constexpr (DCV) {
        ph = cbase / p.phase_c;
        yw -= ph >> 1; x0 -= ph & 1; ylim = p.H;
        c0 -= ph * p.phase_c;
        slot = tile * 4 + ph;
    }
    // PAIR: the two segments are two images (2q, 2q + 1; q = tile / tiles_y): one record set per image, at that image's slot
#pragma unroll
    for (int sg = 0; sg < (PAIR ? 2 : 1); ++sg) {
    if constexpr (PAIR) {
        const int q = tile / p.tiles_y, ty = tile - q * p.tiles_y;
        if (2 * q + sg >= p.N) continue;
        slot = (2 * q + sg) * p.tiles_y + ty;
        if constexpr (DCV) slot = slot * 4 + ph;
    }
    bool ok[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
        ok[m] = PAIR ? ((m & 1) == sg && yw + (m >> 1) < ylim && x0 + li < p.W) : ((yw + (m >> 1) < ylim) && (x0 + li + (m & 1) * 16 < p.W));
    const int rows = min(max(ylim - yw, 0), RPW), cols = min(max(p.W - x0, 0), PAIR ? 16 : 32);
    const float cnt = (float)(rows * cols);                        // valid pixels of this wave (uniform)
    // sums of (x - bias) and of its square per channel: the conv response without its bias has a small mean, so M2 = s2 - s1^2 / n loses nothing
    float s1[16], s2[16];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m) { const float d = ok[m] ? acc[t][m][j] - bias[t][j] : 0.f; a += d; b += d * d; }
            s1[4 * t + j] = a; s2[4 * t + j] = b;
        }
    // transposing reduction over the 16 pixel lanes: at the step of lane bit `bit` a lane keeps the half of its values whose channel index has
    // that bit equal to its own and adds the partner's half -- 8 + 4 + 2 + 1 exchanges instead of 16 x 4; lane li ends with channel li's totals
#define INNFER_TR_STEP(BIT, CNT)                                                                      \
    _Pragma("unroll") for (int i = 0; i < CNT; ++i) {                                                 \
        const bool up = (li & BIT) != 0;                                                              \
        const float k1 = up ? s1[i + CNT] : s1[i], g1 = up ? s1[i] : s1[i + CNT];                     \
        const float k2 = up ? s2[i + CNT] : s2[i], g2 = up ? s2[i] : s2[i + CNT];                     \
        s1[i] = k1 + __shfl_xor(g1, BIT);                                                             \
        s2[i] = k2 + __shfl_xor(g2, BIT);                                                             \
    }
    INNFER_TR_STEP(8, 8)
    INNFER_TR_STEP(4, 4)
    INNFER_TR_STEP(2, 2)
    INNFER_TR_STEP(1, 1)
#undef INNFER_TR_STEP
    // channel of lane li: bit 3 chose between values [0, 8) / [8, 16), bit 2 between the halves of that, ... = value index li = 4 t + j
    float bl = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) bl = (li == 4 * t + j) ? bias[t][j] : bl;
    const float inv = cnt > 0.f ? 1.0f / cnt : 0.f;
    const float mean = s1[0] * inv;
    float* o = p.stats_part + ((long)(slot * NCW + wave) * p.stats_cn + c0 + li) * 3;
    o[0] = cnt; o[1] = bl + mean; o[2] = fmaxf(s2[0] - s1[0] * mean, 0.f);
    }
}

// SGATE (conv3x3_pc<.., TMF | 0x80000>): v = fp16(acc) is the B fragment of the 32 x 32 gate matrix (a lane's 8 accumulators are 8 consecutive channels of its pixel);
// acc <- v * sigmoid(W v + b).  Sigmoid on the hardware exponential / reciprocal: at 2160 x 3840 this epilogue evaluates 265 M of them (the libm forms were 0.3 of the launch).
template <int MT>
__device__ __forceinline__ void self_gate(f32x4 (&acc)[2][MT], const f16x8* sgw, const f32x4* sgb) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        f16x8 vb;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) vb[4 * t + j] = (f16)acc[t][m][j];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 g = __builtin_amdgcn_mfma_f32_16x16x32_f16(sgw[t], vb, sgb[t], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][m][j] = (float)vb[4 * t + j] * __builtin_amdgcn_rcpf(1.0f + __expf(-g[j]));
        }
    }
}

// The same gate in the fp32-accurate mode (SPLIT + SGATE): the accumulators hold the conv's fp32 result; v is split into (hi, lo * 2^11) in registers and
// g = Wh vh + 2^-11 (Wl vh + Wh vl) + b (three MFMAs per 16-row tile), acc <- v * sigmoid(g) with v in fp32.
template <int MT>
__device__ __forceinline__ void self_gate_split(f32x4 (&acc)[2][MT], const f16x8* wh, const f16x8* wl, const f32x4* sgb) {
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        f16x8 vh, vl;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float f = acc[t][m][j];
                const f16 h = (f16)f;
                vh[4 * t + j] = h;
                vl[4 * t + j] = (f16)((f - (float)h) * 2048.0f);
            }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 gm = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], vh, sgb[t], 0, 0, 0);
            f32x4 gx = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t], vh, z4, 0, 0, 0);
            gx = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], vl, gx, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][m][j] = acc[t][m][j] * __builtin_amdgcn_rcpf(1.0f + __expf(-__builtin_fmaf(gx[j], 1.0f / 2048.0f, gm[j])));
        }
    }
}

// RLDS (conv3x3_pc<.., TMF | 0x40000>): the lane's 16 residual channels of each of its MT pixel tiles from the live LDS stage `st` (byte offsets roffs), added to the fp32
// accumulators as x / s1 (rs1 = 1 / s1)
template <int MT>
__device__ __forceinline__ void residual_from_lds(f32x4 (&acc)[4][MT], const char* st, const int* roffs, float rs1) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const f16x8 x0 = *(const f16x8*)(st + roffs[m]), x1 = *(const f16x8*)(st + roffs[m] + 16);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[t][m][j] = __builtin_fmaf((float)(t < 2 ? x0 : x1)[(t & 1) * 4 + j], rs1, acc[t][m][j]);
    }
}

// The same for the plane row order (ROWP): the staged group HALF (0 / 1) holds the residual channels of the lane's tiles 2 half, 2 half + 1 -- 8 channels, one 16-byte slot
template <int MT, int HALF>
__device__ __forceinline__ void residual_from_lds_plane(f32x4 (&acc)[4][MT], const char* st, const int* roffs, float rs1) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const f16x8 x = *(const f16x8*)(st + roffs[m]);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[2 * HALF + tt][m][j] = __builtin_fmaf((float)x[tt * 4 + j], rs1, acc[2 * HALF + tt][m][j]);
    }
}
