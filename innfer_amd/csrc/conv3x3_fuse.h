// FUSE (conv3x3_pc<.., TMF | 0x20000>): HR_conv0 -> conv_last in one kernel -- ring index, pixel store, the fused epilogue and the rim pass.
// Part of csrc/conv3x3.hip (split out in round 5, VERDICT r4 item 7: no functional change -- the device assembly of the translation unit is identical);
// included there, inside namespace innfer { namespace { .. } }, after KP / the tile constants.  Not a stand-alone header.

// FUSE (conv3x3_pc<.., TMF | 0x20000>): index of a pixel of a tile's 18 x 34 neighbourhood (Y in [-1, 16], X in [-1, 32]) that is NOT at least one pixel inside
// the tile, among the 192 such pixels: rows -1, 0 (34 each), rows 15, 16 (34 each), then columns -1, 0, 31, 32 of rows 1 .. 14
__host__ __device__ inline int fuse_ring_index(int Y, int X) {
    if (Y <= 0) return (Y + 1) * 34 + X + 1;
    if (Y >= 15) return 68 + (Y - 15) * 34 + X + 1;
    return 136 + (Y - 1) * 4 + (X <= 0 ? X + 1 : X - 29);
}
constexpr int FUSE_RING = 192;
constexpr int FUSE_PITCH = 516;          // floats between the product planes in LDS (512 pixels + 4: see fused_last_epilogue)

// One finished pixel of the fused last conv: planar fp16 / fp32 [N, oc, H, W] (mode 0 / 1), or the uint8 HWC BGR image of tensor2np (mode 2: the conversion of
// the planar kernel's uint8 epilogue, value for value -- utils.py:197-248)
__device__ __forceinline__ void fuse_store_pixel(void* out, int mode, int denorm, int round16, int oc, long n, int H, int W, int y, int x, const float (&v)[3]) {
    if (mode == 2) {
        uint8_t* o = (uint8_t*)out + ((n * H + y) * W + x) * oc;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (c < oc) {
                float t = round16 ? (float)(f16)v[c] : v[c];
                if (denorm) t = fminf(fmaxf(__fdiv_rn(__fsub_rn(t, -1.0f), 2.0f), 0.0f), 1.0f);
                t = fminf(fmaxf(__fmul_rn(255.0f, t), 0.0f), 255.0f);
                o[oc == 3 ? 2 - c : c] = (uint8_t)__float2int_rn(t);
            }
        return;
    }
    const long plane = (long)H * W, ob = n * oc * plane + (long)y * W + x;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (c < oc) {
            if (mode == 1) ((float*)out)[ob + c * plane] = v[c];
            else ((f16*)out)[ob + c * plane] = (f16)v[c];
        }
}

// The fused last conv (see the FUSE flag of conv3x3_pc) in two halves -- fused_last_epilogue below runs them back to back; csrc/hr_chain.hip puts other work between them.
// fused_last_products: acc = this wave's 2 rows x 32 pixels x 64 channels (bias included) -> activation, fp16, the 27 x 64 last-conv matrix on the matrix cores, the 27 products per
// pixel parked in `pl` (the LDS stage the tile has finished with, >= 64 KB; every wave must be past its last read of it: the caller's barrier); aw: the last conv's four A fragments
// in LDS.  The caller makes the products visible (s_waitcnt lgkmcnt(0) + barrier) before fused_last_sums.
template <int RPW, int NT>
__device__ __forceinline__ void fused_last_products(const KP& p, f32x4 (&acc)[NT][2 * RPW], char* pl, const char* aw, int cw, int lane) {
    constexpr int MT = 2 * RPW;
    const int li = lane & 15, lg = lane >> 4;
    // (three phases with short live ranges -- the kernel's main loop already sits at the 168-register budget: the fp16 fragments of all four pixel tiles first
    //  (the 64 accumulator registers die there), then one A fragment at a time against all of them, then the stores)
    constexpr int KS = NT / 2;                              // 32-channel k-steps: 2 for the 64-channel kernel; 1 for the 32-channel one (round 5: PAN's HRconv + conv_last)
    f16x8 hb[MT][KS];                                       // channels 4 NT lg + 8 ks + e of pixel li: the values the unfused epilogue would have stored
    // (the activation chosen ONCE: a uniform test per value is a branch per value in this unrolled code -- 128 of them cost more than the rest of the epilogue)
    auto to_f16 = [&](auto act_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = acc[2 * ks + (e >> 2)][m][e & 3];
                    if (ACT == 1) f = __builtin_amdgcn_fmed3f(f, 0.2f * f, 3.0e38f);      // (max(f, 0.2 f) as ONE instruction: a finite top keeps LLVM from rewriting it as maxnum + canonicalise)
                    else if (ACT == 2) f = __builtin_amdgcn_fmed3f(f, 0.f, 3.0e38f);
                    FP32_VALUE(f);
                    hb[m][ks][e] = (f16)f;
                }
    };
    if (p.act == 1) to_f16(std::integral_constant<int, 1>{}); else if (p.act == 2) to_f16(std::integral_constant<int, 2>{}); else to_f16(std::integral_constant<int, 0>{});
    f32x4 pa[2][MT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int m = 0; m < MT; ++m) pa[rt][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8 af = *(const f16x8*)(aw + ((rt * KS + ks) * 64 + lane) * 16);
#pragma unroll
            for (int m = 0; m < MT; ++m) pa[rt][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, hb[m][ks], pa[rt][m], 0, 0, 0);
        }
    }
    // 27 planes of 512 pixels, plane pitch 516 floats: a store instruction's 64 lanes (16 pixels x the 4 rows 4 lg + j) and a gather's 64 consecutive
    // pixels of one plane fall into 64 different banks (pixel-major rows of 32 floats were a 32-way conflict)
    char* const plw = pl + ((4 * lg) * FUSE_PITCH + cw * RPW * 32 + li) * 4;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // (64-channel kernel: all 32 rows -- the five padding rows fit its 77-KB stage, a lane-dependent test per store costs more; 32-channel kernel: a 59-KB stage
                //  holds 28 planes, rows 28 .. 31 -- lane group 3 of row tile 1 -- are padding and stay unwritten)
                if (NT == 4 || rt == 0 || lg < 3)
                    *(float*)(plw + ((16 * rt + j) * FUSE_PITCH + (m >> 1) * 32 + (m & 1) * 16) * 4) = pa[rt][m][j];
            }
}

// fused_last_sums: every output pixel of the tile's 18 x 34 neighbourhood sums the nine products that lie inside the tile (see the FUSE flag of conv3x3_pc); tile: the tile's index
// over the batch (n, ty, tx).
template <int RPW>
__device__ __forceinline__ void fused_last_sums(const KP& p, const char* pl, int n, int ty0, int tx0, int cw, int lane, int tile) {
    // Every lane sums its OWN pixel of the tile (512 lanes, 512 pixels: no division, the row tests are uniform but for the first / last wave) ...
    {
        const int Y = cw * RPW + (lane >> 5), X = lane & 31;
        const float* q0 = (const float*)pl + Y * 32 + X;
        float S0 = 0.f, S1 = 0.f, S2 = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const bool ry = dy == 0 ? Y > 0 : (dy == 2 ? Y < 15 : true);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const bool cx = dx == 0 ? X > 0 : (dx == 2 ? X < 31 : true);
                if (ry && cx) {                                   // out(Y, X) += W[dy][dx] . hr(Y + dy - 1, X + dx - 1)
                    const float* q = q0 + (dy * 3 + dx) * 3 * FUSE_PITCH + (dy - 1) * 32 + (dx - 1);
                    S0 += q[0]; S1 += q[FUSE_PITCH]; S2 += q[2 * FUSE_PITCH];
                }
            }
        }
        if (Y >= 1 && Y <= 14 && X >= 1 && X <= 30) {            // complete: every hr pixel it reads lies in this tile
            const float v[3] = {S0 + p.fl_bias[0], p.fl_oc > 1 ? S1 + p.fl_bias[1] : 0.f, p.fl_oc > 2 ? S2 + p.fl_bias[2] : 0.f};
            fuse_store_pixel(p.fl_out, p.fl_out_mode, p.out_denorm, p.out_round16, p.fl_oc, n, p.H, p.W, ty0 + Y, tx0 + X, v);
        } else {
            float* sd = p.fl_side + ((long)tile * FUSE_RING + fuse_ring_index(Y, X)) * 3;
            sd[0] = S0; sd[1] = S1; sd[2] = S2;
        }
    }
    // ... and the first hundred lanes one of the 100 pixels just outside it (rows -1 and 16, columns -1 and 32), which only the tile's edge pixels reach
    const int o = cw * 64 + lane;
    if (o < 100) {
        int Y, X;
        if (o < 34) { Y = -1; X = o - 1; } else if (o < 68) { Y = 16; X = o - 35; } else { Y = (o - 68) >> 1; X = ((o - 68) & 1) ? 32 : -1; }
        float S0 = 0.f, S1 = 0.f, S2 = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int y = Y + dy - 1, x = X + dx - 1;
                if (y >= 0 && y < 16 && x >= 0 && x < 32) {
                    const float* q = (const float*)pl + (dy * 3 + dx) * 3 * FUSE_PITCH + y * 32 + x;
                    S0 += q[0]; S1 += q[FUSE_PITCH]; S2 += q[2 * FUSE_PITCH];
                }
            }
        float* sd = p.fl_side + ((long)tile * FUSE_RING + fuse_ring_index(Y, X)) * 3;
        sd[0] = S0; sd[1] = S1; sd[2] = S2;
    }
}

// acc: this wave's 2 rows x 32 pixels x 64 channels (bias included); pl: the LDS stage the tile has finished with (>= 64 KB); aw: the last conv's four A fragments in LDS; tile: the
// tile's index over the batch (n, ty, tx).
template <int RPW, int NT>
__device__ __forceinline__ void fused_last_epilogue(const KP& p, f32x4 (&acc)[NT][2 * RPW], char* pl, const char* aw, int n, int ty0, int tx0, int cw, int lane, int tile) {
    asm volatile("s_barrier" ::: "memory");                   // every consumer has read its last fragments of this stage: it may be overwritten
    fused_last_products<RPW, NT>(p, acc, pl, aw, cw, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");                   // the tile's products are in LDS
    fused_last_sums<RPW>(p, pl, n, ty0, tx0, cw, lane, tile);
}

// Finishes the rim pixels of the fused last conv: every pixel on the rim of a tile sums, in a fixed order, the partial sums of the tiles whose 18 x 34
// neighbourhood contains it (its own and one to three neighbours), adds the bias and stores.  One thread per (tile, rim pixel): 92 per tile.
__global__ void fuse_combine_kernel(const float* side, const float* bias, void* out, int mode, int denorm, int round16, int oc, int N, int H, int W) {
    const int tiles_x = W / 32, tiles_y = H / 16, per_img = tiles_x * tiles_y;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * per_img * 92) return;
    const int tile = (int)(idx / 92), r = (int)(idx - (long)tile * 92);
    int Y, X;
    if (r < 32) { Y = 0; X = r; } else if (r < 64) { Y = 15; X = r - 32; } else { const int k = r - 64; Y = 1 + (k >> 1); X = (k & 1) ? 31 : 0; }
    const int n = tile / per_img, t = tile - n * per_img, ty = t / tiles_x, tx = t - ty * tiles_x;
    float S[3] = {bias[0], oc > 1 ? bias[1] : 0.f, oc > 2 ? bias[2] : 0.f};
    // the tiles whose 18 x 34 neighbourhood holds this pixel: its own and, for a pixel of the first / last row (column), the tile above / below (left / right) -- in the
    // order of the nine-neighbour scan this replaces (rows of tiles ascending, then columns: the same sums), without its five to eight empty trips per pixel
    const int a0 = Y == 0 ? -1 : 0, a1 = Y == 15 ? 1 : 0, b0 = X == 0 ? -1 : 0, b1 = X == 31 ? 1 : 0;
    for (int a = a0; a <= a1; ++a)
        for (int b = b0; b <= b1; ++b) {
            const int nty = ty + a, ntx = tx + b;
            if (nty < 0 || nty >= tiles_y || ntx < 0 || ntx >= tiles_x) continue;
            const float* sd = side + ((long)(n * per_img + nty * tiles_x + ntx) * FUSE_RING + fuse_ring_index(Y - 16 * a, X - 32 * b)) * 3;
            S[0] += sd[0]; S[1] += sd[1]; S[2] += sd[2];
        }
    fuse_store_pixel(out, mode, denorm, round16, oc, n, H, W, ty * 16 + Y, tx * 32 + X, S);
}
