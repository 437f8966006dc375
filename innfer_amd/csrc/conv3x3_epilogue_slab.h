// Slab epilogues of the conv kernels: plain / canvas / split (fp32-accurate), with the row-order helpers (toff_slab, lane_cbase).
// Part of csrc/conv3x3.hip (split out in round 5, VERDICT r4 item 7: no functional change -- the device assembly of the translation unit is identical);
// included there, inside namespace innfer { namespace { .. } }, after KP / the tile constants.  Not a stand-alone header.

// Offsets of accumulator tile t of a lane from the lane's first channel, in a slab of group stride g (f16 elements) and in a linear channel array (bias), and the lane's
// first channel inside its 16 NT-channel group.  ROWP (conv3x3_pc<.., TMF | 0x400000>, 64-channel groups): the PLANE row order -- accumulator (tile t, row 4 lg + j) is
// channel 32 (t >> 1) + 8 lg + 4 (t & 1) + j instead of 16 lg + 4 t + j: a lane's sixteen channels are 16 bytes in EACH of the group's two 32-channel slab planes, lanes
// lg = 0..3 cover a pixel's whole 64 bytes of one plane, and a store / residual-load instruction touches ONE plane -- half the lines per instruction (measured as an
// ablation first: frame -1.4 %, profiles/r4/upconv_bound.txt).  Panels from conv_pack*(.., rowp = 1).
// LeakyReLU / ReLU as v_med3_f32(f, 0.2 f | 0, top): == f > 0 ? f : 0.2 f | 0 for every |f| <= top, one instruction behind the multiply instead of compare + select.  The top is a FINITE
// constant on purpose: with +inf LLVM rewrites the median as maxnum and, in IEEE mode, puts a canonicalising v_max x, x in front of it (round 6, docs/EXPERIMENTS.md 131).
constexpr float ACT_TOP = 3.0e38f;

template <int NT, bool ROWP>
__device__ __forceinline__ long toff_slab(int t, long g) { return (NT == 4 && ROWP) ? (long)(t >> 1) * g + 4 * (t & 1) : 4 * t; }
template <int NT, bool ROWP>
__device__ __forceinline__ int toff_lin(int t) { return (NT == 4 && ROWP) ? 32 * (t >> 1) + 4 * (t & 1) : 4 * t; }
template <int NT, bool ROWP>
__device__ __forceinline__ int lane_cbase(int lg) { return ((NT == 4 && ROWP) ? 8 : 4 * NT) * lg; }

// Slab epilogue, specialised on (activation, residual 1, residual 2) so that the unrolled loop over the
// wave's pixel tiles is straight-line code: residual loads for all tiles first (their latencies
// overlap), then act -> *s1 + res1 -> *s2 + res2 -> fp16 -> one 8*NT-byte store per pixel tile.
// (The generic runtime-flag version of this loop took ~15 k cycles per workgroup, a third of the
// lifetime of a 64->32 workgroup: profiles/r1/wg_timeline_r1c.txt.)
// POLY: (n, y, x) address a polyphase sub-image of a dilation-d conv: image n / d^2, phase (py, px) = (n % d^2) / d, % d, full-resolution
// pixel (y*d + py, x*d + px); the sub-image ends where the full image does.  No residuals in that mode.
// CV (image canvas, see conv3x3_pc): (ty0, tx0) are canvas coordinates; a pixel tile may lie in the cell below / right of the tile's first
// cell, or on the one-pixel gutter between cells (not stored).  Same arithmetic, per-pixel-tile addresses.
// SC1 (RLDS kernels): res1 is already inside the accumulators as x / s1 (consumer loop); the epilogue only scales by s1 (then R2 as usual).
template <int RPW, int NT, int ACT, bool R1, bool R2, bool SC1 = false, bool ROWP = false>
__device__ __forceinline__ void epilogue_slab_cv(const KP& p, f32x4 (&acc)[NT][2 * RPW], int ty0, int tx0, int wave, int li, int cbase) {
    constexpr int MT = 2 * RPW;
    // ACT 7 (pair gate, PAN's PAConv): the lane's upper NT / 2 channel tiles are the gates of its lower ones -- out = conv_lo * sigmoid(conv_hi),
    // half as many output channels (the launch's rows are ordered so that a value and its gate share a lane)
    constexpr int NTS = ACT == 7 ? NT / 2 : NT;
    const int oc0 = (ACT == 7 ? cbase >> 1 : cbase) + p.out_coff;
    const int cyB = ty0 / p.cv_h1, yB = ty0 - cyB * p.cv_h1 + wave * RPW;        // wave-uniform
    const int cxB = tx0 / p.cv_w1, xB = tx0 - cxB * p.cv_w1 + li;
    f16* ob = (f16*)p.out + (oc0 >> 5) * p.out_gstride + (oc0 & 31);
    const f16* r1b = R1 ? p.res1 + (cbase >> 5) * p.res1_gstride + (cbase & 31) : nullptr;
    const f16* r2b = R2 ? p.res2 + (cbase >> 5) * p.res2_gstride + (cbase & 31) : nullptr;
    bool ok[MT];
    int off[MT];                                   // element offset of the pixel inside a channel group (< 2^31: checked by the host)
    f16x4 r1[R1 ? MT : 1][NT], r2[R2 ? MT : 1][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        int y = yB + (m >> 1), cy = cyB;
        if (y >= p.cv_h1) { y -= p.cv_h1; ++cy; }
        int x = xB + (m & 1) * 16, cx = cxB;
        if (x >= p.cv_w1) { x -= p.cv_w1; ++cx; }
        const int n = cy * p.cv_gx + cx;
        ok[m] = y < p.H && x < p.W && cy < p.cv_gy && cx < p.cv_gx && n < p.N;
        off[m] = ((n * p.H + y) * p.W + x) * 32;
        if (R1 && !R2 && ok[m]) {
#pragma unroll
            for (int t = 0; t < NT; ++t) r1[R1 ? m : 0][t] = *(const f16x4*)(r1b + off[m] + toff_slab<NT, ROWP>(t, p.res1_gstride));
        }
        if (!R1 && R2 && ok[m]) {       // one residual from memory: its loads for all pixel tiles first, like the R1-only form
#pragma unroll
            for (int t = 0; t < NT; ++t) r2[R2 ? m : 0][t] = *(const f16x4*)(r2b + off[m] + toff_slab<NT, ROWP>(t, p.res2_gstride));
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (!ok[m]) continue;
        f16* op = ob + off[m];
        if (R1 && R2) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                r1[R1 ? m : 0][t] = *(const f16x4*)(r1b + off[m] + toff_slab<NT, ROWP>(t, p.res1_gstride));
                r2[R2 ? m : 0][t] = *(const f16x4*)(r2b + off[m] + toff_slab<NT, ROWP>(t, p.res2_gstride));
            }
        }
#pragma unroll
        for (int t = 0; t < NTS; ++t) {
            f16x4 h;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float f = acc[t][m][j];
                if (ACT == 7) f = f * fast_sigmoid(acc[t + NT / 2][m][j]);
                else if (ACT == 1) f = __builtin_amdgcn_fmed3f(f, 0.2f * f, ACT_TOP);       // (max(f, 0.2 f) as ONE instruction; see ACT_TOP)
                else if (ACT == 2) f = __builtin_amdgcn_fmed3f(f, 0.f, ACT_TOP);
                if (SC1) { f = f * p.s1; FP32_VALUE(f); }
                if (R1) f = __builtin_fmaf(f, p.s1, (float)r1[R1 ? m : 0][t][j]);
                if (R2) f = __builtin_fmaf(f, p.s2, (float)r2[R2 ? m : 0][t][j]);
                FP32_VALUE(f);
                h[j] = (f16)f;
            }
            *(f16x4*)(op + toff_slab<NT, ROWP>(t, p.out_gstride)) = h;
        }
    }
}

// DCV (conv3x3_pc<.., TM = 0x1B>: one output phase of ConvTranspose2d(4, 2, 1) per 16*NT-channel group): (ty0, tx0) are coordinates of the phase's
// shifted lattice (see decode); virtual pixel (y', x') of phase (a, b) is output pixel (2y' - a, 2x' - b) of the 2H x 2W slab, channel ch % phase_c.
// PSH (conv3x3_pc<.., TMF | 0x800000>: nn.PixelShuffle(2) as the store, block.py:333-346): the K = 4 * phase_c conv channels arrive PHASE-MAJOR (panels from
// conv_pack_shuffle2: channel ph * phase_c + oc is reference channel 4 oc + ph), so a 64-channel group is one output phase (a, b) = (ph >> 1, ph & 1) of
// 64 consecutive output channels: the DCV store without the lattice shift -- pixel (y, x) of the conv grid goes to (2y + a, 2x + b).
template <int RPW, int NT, int ACT, bool R1, bool R2, bool HOIST, bool POLY = false, bool DCV = false, bool PAIR = false, bool SC1 = false, bool ROWP = false, bool PSH = false>
__device__ __forceinline__ void epilogue_slab(const KP& p, f32x4 (&acc)[NT][2 * RPW], int n, int ty0, int tx0,
                                              int wave, int li, int cbase, int dil = 1) {
    constexpr int MT = 2 * RPW;
    constexpr int NTS = ACT == 7 ? NT / 2 : NT;              // ACT 7: pair gate (see epilogue_slab_cv)
    int oc0 = (ACT == 7 ? cbase >> 1 : cbase) + p.out_coff;
    int yw = ty0 + wave * RPW, xl = tx0 + li;
    long pix0 = ((long)n * p.H + yw) * p.W + xl;
#ifdef INNFER_ABLATE
    if (p.abl & 16) pix0 = (long)blockIdx.x * 64 + wave * RPW * p.W + li;      // every tile of a workgroup stores to the same (cache-resident) lines
#endif
    long rowstep = (long)p.W * 32;
    long colstep = 16 * 32;
    int ylim = p.y1, xlim = p.W;
    if constexpr (POLY) {
        const int d = dil, dd = d * d;
        const int nn = n / dd, ph = n - nn * dd, py = ph / d, px = ph - py * d;
        pix0 = ((long)nn * p.fullH + (long)yw * d + py) * p.fullW + (long)xl * d + px;
        rowstep = (long)p.fullW * 32 * d;
        colstep = 16L * 32 * d;
        ylim = (p.fullH - py + d - 1) / d;
        xlim = (p.fullW - px + d - 1) / d;
    }
    if constexpr (DCV || PSH) {
        const int ph = cbase / p.phase_c, a = ph >> 1, b = ph & 1;
        oc0 -= ph * p.phase_c;
        if constexpr (!PSH) { yw -= a; xl -= b; }       // source pixel of the virtual one (>= 0: the lattice starts at (a, b))
        pix0 = ((long)n * 2 * p.H + 2 * yw + a) * (2 * p.W) + 2 * xl + b;
        rowstep = (long)p.W * 32 * 4;
        colstep = 16 * 32 * 2;
        ylim = p.H;
        if constexpr (PAIR) { pix0 = ((long)(2 * n) * 2 * p.H + 2 * yw + a) * (2 * p.W) + 2 * xl + b; colstep = 4L * p.H * p.W * 32; }
#ifdef INNFER_ABLATE
        // abl 64 (wrong results by construction): phase b of a row goes to the left / right HALF of the HR row as 16 consecutive pixels -- the same bytes as
        // whole 128-byte lines instead of every other 64-byte pixel (what the half-line stores of the phase scatter cost: profiles/r4/upconv_bound.txt)
        if (!PAIR && (p.abl & 64)) { pix0 = ((long)n * 2 * p.H + 2 * yw + a) * (2 * p.W) + xl + b * p.W; colstep = 16 * 32; }
#endif
    } else if constexpr (PAIR) {
        // PAIR (images at most 16 pixels wide): the tile's two 16-pixel segments are images 2n and 2n + 1 -- a segment step is an image step
        pix0 = ((long)(2 * n) * p.H + yw) * p.W + xl;
        colstep = (long)p.H * p.W * 32;
    }
    f16* ob = (f16*)p.out + (oc0 >> 5) * p.out_gstride + pix0 * 32 + (oc0 & 31);
    const f16* r1b = R1 ? p.res1 + (cbase >> 5) * p.res1_gstride + pix0 * 32 + (cbase & 31) : nullptr;
    const f16* r2b = R2 ? p.res2 + (cbase >> 5) * p.res2_gstride + pix0 * 32 + (cbase & 31) : nullptr;
    bool ok[MT];
    f16x4 r1[R1 ? MT : 1][NT], r2[R2 ? MT : 1][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        ok[m] = PAIR ? (2 * n + (m & 1) < p.N && yw + (m >> 1) < ylim && xl < xlim) : ((yw + (m >> 1) < ylim) && (xl + (m & 1) * 16 < xlim));
        const long o = (m >> 1) * rowstep + (m & 1) * colstep;
        if (HOIST && R1 && ok[m]) {
#pragma unroll
            for (int t = 0; t < NT; ++t) r1[R1 ? m : 0][t] = *(const f16x4*)(r1b + o + toff_slab<NT, ROWP>(t, p.res1_gstride));
        }
        if (HOIST && R2 && ok[m]) {
#pragma unroll
            for (int t = 0; t < NT; ++t) r2[R2 ? m : 0][t] = *(const f16x4*)(r2b + o + toff_slab<NT, ROWP>(t, p.res2_gstride));
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (!ok[m]) continue;
#ifdef INNFER_ABLATE
        if (p.abl & 1) continue;
#endif
        f16* op = ob + (m >> 1) * rowstep + (m & 1) * colstep;
#ifdef INNFER_ABLATE
        // abl 128 (wrong results by construction): every lane's 32 bytes of pixel tile m at consecutive addresses -- a store instruction writes 1 KB of whole
        // lines instead of 16-byte pieces of 32 (what the piece-wise stores of the MFMA result layout cost)
        if (DCV && !PAIR && (p.abl & 128)) op = (f16*)p.out + (pix0 - 2 * li) * 32 + (m * 64 + (int)(threadIdx.x & 63)) * 16;
        // abl 256 (values land permuted inside the wave's own 1 KB runs; same bytes, same lines, no overlap between waves): lane L writes piece L of the run --
        // consecutive lanes -> consecutive addresses -- instead of lane (li, lg) -> pixel li, piece lg: what the LANE ORDER of the MFMA result layout costs
        if (!DCV && !PAIR && !POLY && (p.abl & 256)) {
            const int lgq = (int)(threadIdx.x & 63) >> 4;
            if (NT == 2) op = op - li * 32 - 8 * lgq + (int)(threadIdx.x & 63) * 8;
            else if (NT == 4) op = op - li * 32 - 16 * (lgq & 1) + ((lgq & 1) * 16 + li) * 16;
        }
        if (!DCV && !PAIR && !POLY && (p.abl & 128)) op = (f16*)p.out + (pix0 - li) * 32 + (m * 64 + (int)(threadIdx.x & 63)) * (4 * NT);      // (the plain layers: 16 / 32 bytes per lane)
#endif
        if (!HOIST) {
            const long o = (m >> 1) * rowstep + (m & 1) * colstep;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (R1) r1[R1 ? m : 0][t] = *(const f16x4*)(r1b + o + toff_slab<NT, ROWP>(t, p.res1_gstride));
                if (R2) r2[R2 ? m : 0][t] = *(const f16x4*)(r2b + o + toff_slab<NT, ROWP>(t, p.res2_gstride));
            }
        }
#pragma unroll
        for (int t = 0; t < NTS; ++t) {
            f16x4 h;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float f = acc[t][m][j];
                if (ACT == 7) {
                    f = f * fast_sigmoid(acc[t + NT / 2][m][j]);
                } else if (ACT >= 4) {           // pixel-attention gate (PAN): res1 * sigmoid(conv), ACT 4: LeakyReLU(0.2) after it
                    f = (float)r1[R1 ? m : 0][t][j] * (fast_sigmoid(f));
                    if (ACT == 4) f = fmaxf(f, 0.2f * f);
                } else {
                    if (ACT == 1) f = __builtin_amdgcn_fmed3f(f, 0.2f * f, ACT_TOP);
                    else if (ACT == 2) f = __builtin_amdgcn_fmed3f(f, 0.f, ACT_TOP);
                    if (SC1) { f = f * p.s1; FP32_VALUE(f); }          // RLDS: res1 is inside the accumulator as x / s1
                    if (R1) f = __builtin_fmaf(f, p.s1, (float)r1[R1 ? m : 0][t][j]);
                    if (R2) f = __builtin_fmaf(f, p.s2, (float)r2[R2 ? m : 0][t][j]);
                }
                FP32_VALUE(f);
                h[j] = (f16)f;
            }
            *(f16x4*)(op + toff_slab<NT, ROWP>(t, p.out_gstride)) = h;
        }
    }
}

// SPLIT (fp32-accurate mode, conv3x3_pc<.., TMF | 0x2000>): a tensor is a PAIR of fp16 slabs -- hi = fp16(x) and lo = fp16((x - hi) * 2^11), the lo slab a
// fixed distance behind the hi slab -- i.e. 22 significant bits per value with the fp16 kernels' data path.  The epilogue works on the fp32
// accumulators exactly like the fp16 one (activation, *s1 + res1, *s2 + res2 with explicit fmaf) but reads its residuals as hi + lo * 2^-11 (exact
// in fp32) and stores both parts.  CV: image-canvas addressing (see epilogue_slab_cv).
constexpr float SPLIT_UP = 2048.0f, SPLIT_DOWN = 1.0f / 2048.0f;
template <int RPW, int NT, int ACT, bool R1, bool R2, bool CV>
__device__ __forceinline__ void epilogue_slab_split(const KP& p, f32x4 (&acc)[NT][2 * RPW], int n, int ty0, int tx0, int wave, int li, int cbase) {
    constexpr int MT = 2 * RPW;
    const int oc0 = cbase + p.out_coff;
    f16* ob = (f16*)p.out + (oc0 >> 5) * p.out_gstride + (oc0 & 31);
    const f16* r1b = R1 ? p.res1 + (cbase >> 5) * p.res1_gstride + (cbase & 31) : nullptr;
    const f16* r2b = R2 ? p.res2 + (cbase >> 5) * p.res2_gstride + (cbase & 31) : nullptr;
    int cyB = 0, yB = 0, cxB = 0, xB = 0;
    if constexpr (CV) {
        cyB = ty0 / p.cv_h1; yB = ty0 - cyB * p.cv_h1 + wave * RPW;
        cxB = tx0 / p.cv_w1; xB = tx0 - cxB * p.cv_w1 + li;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        bool ok;
        long off;                                  // element offset of the pixel inside a channel group
        if constexpr (CV) {
            int y = yB + (m >> 1), cy = cyB;
            if (y >= p.cv_h1) { y -= p.cv_h1; ++cy; }
            int x = xB + (m & 1) * 16, cx = cxB;
            if (x >= p.cv_w1) { x -= p.cv_w1; ++cx; }
            const int nn = cy * p.cv_gx + cx;
            ok = y < p.H && x < p.W && cy < p.cv_gy && cx < p.cv_gx && nn < p.N;
            off = (((long)nn * p.H + y) * p.W + x) * 32;
        } else {
            const int y = ty0 + wave * RPW + (m >> 1), x = tx0 + li + (m & 1) * 16;
            ok = y < p.y1 && x < p.W;
            off = (((long)n * p.H + y) * p.W + x) * 32;
        }
        if (!ok) continue;
        f16x4 r1h[NT], r1l[NT], r2h[NT], r2l[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (R1) { r1h[t] = *(const f16x4*)(r1b + off + 4 * t); r1l[t] = *(const f16x4*)(r1b + p.res1_lo + off + 4 * t); }
            if (R2) { r2h[t] = *(const f16x4*)(r2b + off + 4 * t); r2l[t] = *(const f16x4*)(r2b + p.res2_lo + off + 4 * t); }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f16x4 h, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float f = acc[t][m][j];
                if (ACT >= 4) {                  // pixel-attention gate (PAN, fp32 mode): res1 * sigmoid(conv), ACT 4: LeakyReLU(0.2) after it
                    f = __builtin_fmaf((float)r1l[t][j], SPLIT_DOWN, (float)r1h[t][j]) * fast_sigmoid(f);
                    if (ACT == 4) f = fmaxf(f, 0.2f * f);
                } else {
                    if (ACT == 1) f = __builtin_amdgcn_fmed3f(f, 0.2f * f, ACT_TOP);
                    else if (ACT == 2) f = __builtin_amdgcn_fmed3f(f, 0.f, ACT_TOP);
                    if (R1) f = __builtin_fmaf(f, p.s1, __builtin_fmaf((float)r1l[t][j], SPLIT_DOWN, (float)r1h[t][j]));
                    if (R2) f = __builtin_fmaf(f, p.s2, __builtin_fmaf((float)r2l[t][j], SPLIT_DOWN, (float)r2h[t][j]));
                }
                FP32_VALUE(f);
                h[j] = (f16)f;
                float d = f - (float)h[j];
                FP32_VALUE(d);
                l[j] = (f16)(d * SPLIT_UP);
            }
            *(f16x4*)(ob + off + 4 * t) = h;
            *(f16x4*)(ob + p.out_lo + off + 4 * t) = l;
        }
    }
}
