// HBM-bound pointwise / gather kernels around the conv stack:
//   - NCHW <-> fp16 blocked-NHWC slab (test helpers of the single-conv entry point)
//   - chop_forward tile extraction  (utils/utils.py:318-369 as used by run.py:178-181)
//   - overlap blend / recompose     (utils/utils.py:372-445)
//   - uint8 HWC BGR <-> float NCHW RGB pre/post (utils/utils.py:164-194,197-248)
// All of them are one read + one write per element; the blend is written as a
// GATHER (one thread per output pixel walks the <=3x3 tiles covering it in the
// reference's (h,w) order) so there are no atomics, no read-modify-write of the
// 3.2 GB output and the fp32 result is bit-identical to the reference's
// scatter loop.
#include "common.h"

// Bit-exact parity with the reference's separate torch ops (mul, then +=, then /) needs every
// rounding kept: hipcc defaults to -ffp-contract=fast, which would fuse a*w + num into one FMA.
#pragma clang fp contract(off)

namespace innfer {
namespace {

__global__ void k_nchw_to_slab(const void* src, int f32, f16* slab, long gstride, int ch_off, int C, long hw, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // over N*C*H*W, pixel fastest
    if (i >= total) return;
    const long px = i % hw;
    const int c = (int)((i / hw) % C);
    const long n = i / (hw * C);
    const float v = f32 ? ((const float*)src)[i] : (float)((const f16*)src)[i];
    const int ch = ch_off + c;
    slab[(ch >> 5) * gstride + (n * hw + px) * 32 + (ch & 31)] = (f16)v;
}

__global__ void k_slab_to_nchw(const f16* slab, long gstride, int ch_off, void* dst, int f32, int C, long hw, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long px = i % hw;
    const int c = (int)((i / hw) % C);
    const long n = i / (hw * C);
    const int ch = ch_off + c;
    const f16 v = slab[(ch >> 5) * gstride + (n * hw + px) * 32 + (ch & 31)];
    if (f32) ((float*)dst)[i] = (float)v; else ((f16*)dst)[i] = v;
}

// (hi, lo) slab pairs of the fp32-accurate mode (conv3x3.hip SPLIT): hi = fp16(x), lo = fp16((x - hi) * 2^11)
__global__ void k_nchw_to_slab_split(const float* src, f16* slab, long gstride, long lo, int ch_off, int C, long hw, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long px = i % hw;
    const int c = (int)((i / hw) % C);
    const long n = i / (hw * C);
    const float v = src[i];
    const int ch = ch_off + c;
    const long o = (ch >> 5) * gstride + (n * hw + px) * 32 + (ch & 31);
    const f16 h = (f16)v;
    slab[o] = h;
    slab[o + lo] = (f16)((v - (float)h) * 2048.0f);
}

__global__ void k_slab_split_to_nchw(const f16* slab, long gstride, long lo, int ch_off, float* dst, int C, long hw, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long px = i % hw;
    const int c = (int)((i / hw) % C);
    const long n = i / (hw * C);
    const int ch = ch_off + c;
    const long o = (ch >> 5) * gstride + (n * hw + px) * 32 + (ch & 31);
    dst[i] = __builtin_fmaf((float)slab[o + lo], 1.0f / 2048.0f, (float)slab[o]);
}

// Four consecutive pixels per thread: when P, the tile step and the output width are multiples of 4 every tile origin is, so the four pixels lie
// under the same tiles and every tile row is read with one 8- / 16-byte load and written with one store.  The ramp step of torch.linspace (one
// IEEE division) is computed once per thread, not per weight.  Same products, sums and division per pixel in the same order as k_recompose
// (bit-identical; the launcher picks this form whenever the geometry allows).
template <typename TI, typename TO>
__global__ void k_recompose_x4(const TI* tiles, TO* out, int C, int P, int FH, int FW, int eff, int nh, int nw, int ov);

template <typename T>
__global__ void k_extract(const T* img, T* tiles, int C, int H, int W, int ps, int step_int,
                          int nw, int tile_begin, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // over count*C*ps*ps
    if (i >= total) return;
    const int x = (int)(i % ps);
    const int y = (int)((i / ps) % ps);
    const int c = (int)((i / ((long)ps * ps)) % C);
    const int k = (int)(i / ((long)ps * ps * C)) + tile_begin;
    const int th = k / nw, tw = k % nw;
    int oy = th * step_int; if (oy > H - ps) oy = H - ps;      // ragged last row anchored at H-ps
    int ox = tw * step_int; if (ox > W - ps) ox = W - ps;
    tiles[i] = img[((long)c * H + oy + y) * W + ox + x];
}

// four consecutive pixels per thread (8- / 16-byte copies): patch size, tile step and image width multiples of 4
template <typename T>
__global__ void k_extract_x4(const T* img, T* tiles, int C, int H, int W, int ps, int step_int, int nw, int tile_begin, long total4) {
    typedef T v4 __attribute__((ext_vector_type(4)));
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // over count*C*ps*(ps/4)
    if (i >= total4) return;
    const int q = ps >> 2;
    const int x = (int)(i % q) * 4;
    const int y = (int)((i / q) % ps);
    const int c = (int)((i / ((long)q * ps)) % C);
    const int k = (int)(i / ((long)q * ps * C)) + tile_begin;
    const int th = k / nw, tw = k % nw;
    int oy = th * step_int; if (oy > H - ps) oy = H - ps;
    int ox = tw * step_int; if (ox > W - ps) ox = W - ps;
    *(v4*)(tiles + i * 4) = *(const v4*)(img + ((long)c * H + oy + y) * W + ox + x);
}

// extract_patches_2d of np2tensor(img) without the float image in between: tile element = float32(u8) / 255 [-> (x - 0.5) * 2 clamped]
// [-> fp16], channels flipped BGR -> RGB (3n channels: full flip; 4: [2,1,0,3]) -- the values k_u8_to_nchw + k_extract produce
template <typename TO>
__global__ void k_extract_u8(const uint8_t* img, TO* tiles, int C, int H, int W, int ps, int step_int, int nw, int tile_begin, long total,
                             int normalize) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // over count*C*ps*ps
    if (i >= total) return;
    const int x = (int)(i % ps);
    const int y = (int)((i / ps) % ps);
    const int c = (int)((i / ((long)ps * ps)) % C);
    const int k = (int)(i / ((long)ps * ps * C)) + tile_begin;
    const int th = k / nw, tw = k % nw;
    int oy = th * step_int; if (oy > H - ps) oy = H - ps;
    int ox = tw * step_int; if (ox > W - ps) ox = W - ps;
    int sc = c;
    if (C % 3 == 0) sc = C - 1 - c; else if (C == 4 && c < 3) sc = 2 - c;
    float v = __fdiv_rn((float)img[((long)(oy + y) * W + ox + x) * C + sc], 255.0f);
    if (normalize) v = fminf(fmaxf(__fmul_rn(__fsub_rn(v, 0.5f), 2.0f), -1.0f), 1.0f);
    tiles[i] = (TO)v;
}

// torch.linspace(start, end, steps)[i] in fp32 = one fused multiply-add per element,
// counted from the nearer end (ATen RangeFactories; verified against golden G2).
__device__ __forceinline__ float lin(float start, float end, int steps, int i) {
    if (steps == 1) return start;
    const float step = __fdiv_rn(__fsub_rn(end, start), (float)(steps - 1));
    return i < steps / 2 ? __fmaf_rn(step, (float)i, start)
                         : __fmaf_rn(-step, (float)(steps - i - 1), end);
}

__device__ __forceinline__ float profile(int i, int P, int ov) {
    if (i < ov) return lin(0.1f, 1.0f, ov, i);
    if (i < P - ov) return 1.0f;
    return lin(1.0f, 0.1f, ov, i - (P - ov));
}

// U8OUT: tensor2np of the blended image as the store -- the blended value is rounded to TO (the tensor recompose_tensor would have returned),
// optionally denormalised, scaled, clipped, rounded half to even and written as a uint8 HWC BGR(A) pixel (`img`, batch 1).
template <typename TI, typename TO, bool U8OUT = false>
__global__ void k_recompose(const TI* tiles, TO* out, int C, int P, int FH, int FW, int eff,
                            int nh, int nw, int ov, uint8_t* img = nullptr, int denormalize = 0) {
    const int X = blockIdx.x * blockDim.x + threadIdx.x;
    const int Y = blockIdx.y;
    const int b = blockIdx.z;
    if (X >= FW) return;
    float num[4] = {0.f, 0.f, 0.f, 0.f};
    float den = 0.f;
    // first lattice tile whose span [h*eff, h*eff+P) can reach Y (clamped origins only move up)
    const int h0 = max(0, (Y - P + eff) / eff), w0 = max(0, (X - P + eff) / eff);
    for (int cb = 0; cb < C; cb += 4) {
        den = 0.f;
        num[0] = num[1] = num[2] = num[3] = 0.f;
        for (int h = h0; h < nh; ++h) {
            const int oy = min(h * eff, FH - P);
            if (oy > Y) break;
            if (Y - oy >= P) continue;
            const float wy = profile(Y - oy, P, ov);
            for (int w = w0; w < nw; ++w) {
                const int ox = min(w * eff, FW - P);
                if (ox > X) break;
                if (X - ox >= P) continue;
                const float wgt = __fmul_rn(profile(X - ox, P, ov), wy);
                den = __fadd_rn(den, wgt);
                const long k = ((long)b * nh + h) * nw + w;
                const TI* tp = tiles + ((k * C + cb) * P + (Y - oy)) * (long)P + (X - ox);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (cb + c < C)
                        num[c] = __fadd_rn(num[c], __fmul_rn((float)tp[(long)c * P * P], wgt));
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (cb + c < C) {
                const TO r = (TO)__fdiv_rn(num[c], den);
                if constexpr (U8OUT) {
                    float v = (float)r;
                    if (denormalize) v = fminf(fmaxf(__fdiv_rn(__fsub_rn(v, -1.0f), 2.0f), 0.0f), 1.0f);
                    v = fminf(fmaxf(__fmul_rn(255.0f, v), 0.0f), 255.0f);
                    const int ch = cb + c, sc = (C == 3 || (C == 4 && ch < 3)) ? 2 - ch : ch;
                    img[((long)Y * FW + X) * C + sc] = (uint8_t)__float2int_rn(v);
                } else {
                    out[(((long)b * C + cb + c) * FH + Y) * FW + X] = r;
                }
            }
    }
}

template <typename TI, typename TO>
__global__ void k_recompose_x4(const TI* tiles, TO* out, int C, int P, int FH, int FW, int eff, int nh, int nw, int ov) {
    typedef TI vin __attribute__((ext_vector_type(4)));
    typedef TO vout __attribute__((ext_vector_type(4)));
    const int X = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int Y = blockIdx.y;
    const int b = blockIdx.z;
    if (X >= FW) return;
    const float st = ov > 1 ? __fdiv_rn(__fsub_rn(1.0f, 0.1f), (float)(ov - 1)) : 0.f;     // linspace(0.1, 1, ov) step; the falling ramp's is -st exactly
    auto prof = [&](int i) -> float {                       // == profile(i, P, ov)
        if (i < ov) return ov == 1 ? 0.1f : (i < ov / 2 ? __fmaf_rn(st, (float)i, 0.1f) : __fmaf_rn(-st, (float)(ov - i - 1), 1.0f));
        if (i < P - ov) return 1.0f;
        const int k = i - (P - ov);
        return ov == 1 ? 1.0f : (k < ov / 2 ? __fmaf_rn(-st, (float)k, 1.0f) : __fmaf_rn(st, (float)(ov - k - 1), 0.1f));
    };
    const int h0 = max(0, (Y - P + eff) / eff), w0 = max(0, (X - P + eff) / eff);
    for (int cb = 0; cb < C; cb += 4) {                     // channels in groups of four: one set of weights per group
        float num[4][4], den[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) num[c][j] = 0.f;
        for (int h = h0; h < nh; ++h) {
            const int oy = min(h * eff, FH - P);
            if (oy > Y) break;
            if (Y - oy >= P) continue;
            const float wy = prof(Y - oy);
            for (int w = w0; w < nw; ++w) {
                const int ox = min(w * eff, FW - P);
                if (ox > X) break;
                if (X - ox >= P) continue;
                const long k = ((long)b * nh + h) * nw + w;
                float wgt[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    wgt[j] = __fmul_rn(prof(X + j - ox), wy);
                    den[j] = __fadd_rn(den[j], wgt[j]);
                }
                const TI* tp = tiles + ((k * C + cb) * P + (Y - oy)) * (long)P + (X - ox);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (cb + c < C) {
                        const vin v = *(const vin*)(tp + (long)c * P * P);
#pragma unroll
                        for (int j = 0; j < 4; ++j) num[c][j] = __fadd_rn(num[c][j], __fmul_rn((float)v[j], wgt[j]));
                    }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (cb + c < C) {
                vout o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (TO)__fdiv_rn(num[c][j], den[j]);
                *(vout*)(out + (((long)b * C + cb + c) * FH + Y) * FW + X) = o;
            }
    }
}

// np2tensor: float32(u8)/255 -> HWC->CHW -> BGR->RGB flip (C%3==0: full flip; C==4: [2,1,0,3])
// -> optional ((x-0.5)*2).clamp(-1,1).  One thread per pixel, all channels.
// TI uint8_t / uint16_t, maxval = MAX_VALUES_BY_DTYPE (utils.py:22-33: 255 / 65535; 1 with change_range=False); flip 0: bgr2rgb=False
template <typename TI, typename TO>
__global__ void k_u8_to_nchw(const TI* img, TO* out, long hw, int C, int normalize, float maxval, int flip) {
    const long px = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= hw) return;
    for (int c = 0; c < C; ++c) {
        int sc = c;
        if (flip) { if (C % 3 == 0) sc = C - 1 - c; else if (C == 4 && c < 3) sc = 2 - c; }
        float v = __fdiv_rn((float)img[px * C + sc], maxval);
        if (normalize) {
            v = __fmul_rn(__fsub_rn(v, 0.5f), 2.0f);
            v = fminf(fmaxf(v, -1.0f), 1.0f);
        }
        out[(long)c * hw + px] = (TO)v;
    }
}

// tensor2np: .float() -> RGB->BGR flip -> CHW->HWC -> optional denorm ((x+1)/2).clip(0,1)
// -> clip(255*x, 0, 255).round() (half to even) -> uint8.
// TU uint8_t / uint16_t with data_range 255 / 65535 (tensor2np(data_range=, imtype=)); flip 0: rgb2bgr=False
template <typename TI, typename TU>
__global__ void k_nchw_to_u8(const TI* in, TU* img, long hw, int C, int denormalize, float range, int flip) {
    const long px = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= hw) return;
    for (int c = 0; c < C; ++c) {
        int sc = c;
        if (flip) { if (C == 3) sc = 2 - c; else if (C == 4 && c < 3) sc = 2 - c; }
        float v = (float)in[(long)sc * hw + px];
        if (denormalize) {
            v = __fdiv_rn(__fsub_rn(v, -1.0f), 2.0f);
            v = fminf(fmaxf(v, 0.0f), 1.0f);
        }
        v = __fmul_rn(range, v);
        v = fminf(fmaxf(v, 0.0f), range);
        img[px * C + c] = (TU)__float2int_rn(v);               // round half to even
    }
}

// colors.py:29-46: linear = float32(srgb)/255; linear <= 0.04045 ? linear/12.92 : ((linear+0.055)/1.055)^2.4
__global__ void k_srgb2linear(const uint8_t* in, float* out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float l = __fdiv_rn((float)in[i], 255.0f);
    out[i] = l <= 0.04045f ? __fdiv_rn(l, 12.92f) : powf(__fdiv_rn(__fadd_rn(l, 0.055f), 1.055f), 2.4f);
}

// colors.py:49-60: clip to [0,1]; s <= 0.0031308 ? s*12.92 : 1.055*s^(1/2.4) - 0.055; clip(s*255, 0, 255) TRUNCATED to uint8
__global__ void k_linear2srgb(const float* in, uint8_t* out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = fminf(fmaxf(in[i], 0.0f), 1.0f);
    s = s <= 0.0031308f ? __fmul_rn(s, 12.92f) : __fsub_rn(__fmul_rn(1.055f, powf(s, (float)(1.0 / 2.4))), 0.055f);
    s = fminf(fmaxf(__fmul_rn(s, 255.0f), 0.0f), 255.0f);
    out[i] = (uint8_t)(int)s;                                   // astype(np.uint8): truncation
}

inline unsigned blocks(long total, int bs) { return (unsigned)((total + bs - 1) / bs); }

}  // namespace

int nchw_to_slab(const void* src, int f32, f16* slab, long gstride, int ch_off, int N, int C, int H, int W, hipStream_t s) {
    const long hw = (long)H * W, total = hw * C * N;
    if (total == 0) return INNFER_OK;
    hipLaunchKernelGGL(k_nchw_to_slab, dim3(blocks(total, 256)), dim3(256), 0, s, src, f32, slab, gstride, ch_off, C, hw, total);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

int slab_to_nchw(const f16* slab, long gstride, int ch_off, void* dst, int f32, int N, int C, int H, int W, hipStream_t s) {
    const long hw = (long)H * W, total = hw * C * N;
    if (total == 0) return INNFER_OK;
    hipLaunchKernelGGL(k_slab_to_nchw, dim3(blocks(total, 256)), dim3(256), 0, s, slab, gstride, ch_off, dst, f32, C, hw, total);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

}  // namespace innfer

using namespace innfer;

extern "C" int innfer_nchw_to_slab_split(const float* d_src, void* d_slab, int64_t group_stride, int64_t lo, int ch_off, int N, int C, int H, int W, void* stream) {
    if (!d_src || !d_slab || lo <= 0) return set_error(INNFER_ERR_INVALID, "nchw_to_slab_split: null argument / lo <= 0");
    const long hw = (long)H * W, total = hw * C * N;
    if (total == 0) return INNFER_OK;
    hipLaunchKernelGGL(k_nchw_to_slab_split, dim3(blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, d_src, (f16*)d_slab, (long)group_stride, (long)lo, ch_off, C, hw, total);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

extern "C" int innfer_slab_split_to_nchw(const void* d_slab, int64_t group_stride, int64_t lo, int ch_off, float* d_dst, int N, int C, int H, int W, void* stream) {
    if (!d_dst || !d_slab || lo <= 0) return set_error(INNFER_ERR_INVALID, "slab_split_to_nchw: null argument / lo <= 0");
    const long hw = (long)H * W, total = hw * C * N;
    if (total == 0) return INNFER_OK;
    hipLaunchKernelGGL(k_slab_split_to_nchw, dim3(blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, (const f16*)d_slab, (long)group_stride, (long)lo, ch_off, d_dst, C, hw, total);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

// ------------------------------------------------------------------ C ABI part
static int axis_plan(int size, int ps, int step_int, int* org) {
    int n = (size - ps) / step_int + 1;
    if (org) for (int i = 0; i < n; ++i) org[i] = i * step_int;
    if ((size - ps) % step_int != 0) { if (org) org[n] = size - ps; ++n; }
    return n;
}

extern "C" int innfer_chop_plan(int H, int W, int patch, double step, int* ps_out, int* nh, int* nw,
                                int* ys, int* xs) {
    if (H <= 0 || W <= 0 || patch <= 0 || !(step > 0.0))
        return set_error(INNFER_ERR_INVALID, "chop_plan: bad arguments H=%d W=%d patch=%d step=%g", H, W, patch, step);
    int ps = patch; if (H < ps) ps = H; if (W < ps) ps = W;          // run.py:176
    const int step_int = (int)(ps * step);                             // utils.py:351-352
    if (step_int <= 0) return set_error(INNFER_ERR_INVALID, "chop_plan: step too small");
    const int a = axis_plan(H, ps, step_int, ys), b = axis_plan(W, ps, step_int, xs);
    if (ps_out) *ps_out = ps;
    if (nh) *nh = a;
    if (nw) *nw = b;
    return INNFER_OK;
}

extern "C" int innfer_extract_tiles(const void* d_img, int dtype, int C, int H, int W, int patch, double step,
                                    int tile_begin, int tile_count, void* d_tiles, void* stream) {
    int ps, nh, nw;
    int rc = innfer_chop_plan(H, W, patch, step, &ps, &nh, &nw, nullptr, nullptr);
    if (rc) return rc;
    if (tile_begin < 0 || tile_count < 0 || tile_begin + tile_count > nh * nw)
        return set_error(INNFER_ERR_INVALID, "extract_tiles: range [%d,+%d) outside %d tiles", tile_begin, tile_count, nh * nw);
    const long total = (long)tile_count * C * ps * ps;
    if (total == 0) return INNFER_OK;
    const int step_int = (int)(ps * step);
    hipStream_t s = (hipStream_t)stream;
    const bool x4 = ps % 4 == 0 && step_int % 4 == 0 && W % 4 == 0;
    if (x4 && dtype == INNFER_F16)
        hipLaunchKernelGGL(k_extract_x4<f16>, dim3(blocks(total / 4, 256)), dim3(256), 0, s, (const f16*)d_img, (f16*)d_tiles,
                           C, H, W, ps, step_int, nw, tile_begin, total / 4);
    else if (x4 && dtype == INNFER_F32)
        hipLaunchKernelGGL(k_extract_x4<float>, dim3(blocks(total / 4, 256)), dim3(256), 0, s, (const float*)d_img, (float*)d_tiles,
                           C, H, W, ps, step_int, nw, tile_begin, total / 4);
    else if (dtype == INNFER_F16)
        hipLaunchKernelGGL(k_extract<f16>, dim3(blocks(total, 256)), dim3(256), 0, s, (const f16*)d_img, (f16*)d_tiles,
                           C, H, W, ps, step_int, nw, tile_begin, total);
    else if (dtype == INNFER_F32)
        hipLaunchKernelGGL(k_extract<float>, dim3(blocks(total, 256)), dim3(256), 0, s, (const float*)d_img, (float*)d_tiles,
                           C, H, W, ps, step_int, nw, tile_begin, total);
    else return set_error(INNFER_ERR_INVALID, "extract_tiles: bad dtype %d", dtype);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

// overlap = scale * int(round((1-step) * (P/scale)))  with Python's round-half-to-even (utils.py:396)
static int blend_overlap(int P, double step, int scale) {
    const double v = (1.0 - step) * ((double)P / scale);
    double r = __builtin_rint(v);                                     // FE_TONEAREST = half to even
    return scale * (int)r;
}

static float lin_host(float start, float end, int steps, int i) {
    if (steps == 1) return start;
    const float step = (end - start) / (float)(steps - 1);
    return i < steps / 2 ? __builtin_fmaf(step, (float)i, start)
                         : __builtin_fmaf(-step, (float)(steps - i - 1), end);
}

extern "C" int innfer_blend_profile(int P, double step, int scale, float* h_profile) {
    if (P <= 0 || scale <= 0 || step < 0.5 || step > 1.0 || !h_profile)
        return set_error(INNFER_ERR_INVALID, "blend_profile: bad arguments");
    const int ov = blend_overlap(P, step, scale);
    if (P - 2 * ov < 0)
        return set_error(INNFER_ERR_INVALID, "blend_profile: overlap %d exceeds half of patch %d "
                         "(the reference raises here: torch.ones(negative), utils.py:415)", ov, P);
    for (int i = 0; i < P; ++i)
        h_profile[i] = i < ov ? lin_host(0.1f, 1.0f, ov, i)
                     : (i < P - ov ? 1.0f : lin_host(1.0f, 0.1f, ov, i - (P - ov)));
    return INNFER_OK;
}

extern "C" int innfer_recompose(const void* d_tiles, int dtype, int n, int C, int P, int height, int width,
                                double step, int scale, void* d_out, int out_dtype, void* stream) {
    if (step < 0.5 || step > 1.0) return set_error(INNFER_ERR_INVALID, "recompose: step must be in [0.5,1]");
    if (n <= 0 || C <= 0 || P <= 0 || scale <= 0) return set_error(INNFER_ERR_INVALID, "recompose: bad sizes");
    const int FH = scale * height, FW = scale * width;
    if (FH < P || FW < P) return set_error(INNFER_ERR_INVALID, "recompose: patch %d larger than output %dx%d", P, FH, FW);
    const int ov = blend_overlap(P, step, scale);
    if (P - 2 * ov < 0)
        return set_error(INNFER_ERR_INVALID, "recompose: overlap %d exceeds half of patch %d (reference raises too)", ov, P);
    const int eff = (int)(step * P), step_int = (int)(P * step);
    const int nh = 1 + (FH - P) / step_int + ((FH - P) % step_int != 0);
    const int nw = 1 + (FW - P) / step_int + ((FW - P) % step_int != 0);
    if (n % (nh * nw)) return set_error(INNFER_ERR_INVALID, "recompose: %d tiles is not a multiple of %dx%d", n, nh, nw);
    const int nb = n / (nh * nw);
    hipStream_t s = (hipStream_t)stream;
    // four pixels per thread when every tile origin (multiples of eff, or FW - P for the ragged last column) and the row pitch are multiples of 4
    const bool x4 = P % 4 == 0 && eff % 4 == 0 && FW % 4 == 0;
    dim3 grid(x4 ? (FW / 4 + 127) / 128 : (FW + 255) / 256, FH, nb), block(x4 ? 128 : 256);
#define RC(TI, TO) do { if (x4) hipLaunchKernelGGL((k_recompose_x4<TI, TO>), grid, block, 0, s, (const TI*)d_tiles, (TO*)d_out, C, P, FH, FW, eff, nh, nw, ov); \
                        else hipLaunchKernelGGL((k_recompose<TI, TO>), grid, block, 0, s, (const TI*)d_tiles, (TO*)d_out, C, P, FH, FW, eff, nh, nw, ov); } while (0)
    if (dtype == INNFER_F16 && out_dtype == INNFER_F16) RC(f16, f16);
    else if (dtype == INNFER_F16 && out_dtype == INNFER_F32) RC(f16, float);
    else if (dtype == INNFER_F32 && out_dtype == INNFER_F32) RC(float, float);
    else if (dtype == INNFER_F32 && out_dtype == INNFER_F16) RC(float, f16);
    else return set_error(INNFER_ERR_INVALID, "recompose: bad dtype");
#undef RC
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

extern "C" int innfer_extract_tiles_u8(const uint8_t* d_img, int C, int H, int W, int normalize, int patch, double step,
                                       int tile_begin, int tile_count, void* d_tiles, int tile_dtype, void* stream) {
    if (!d_img || !d_tiles || C <= 0) return set_error(INNFER_ERR_INVALID, "extract_tiles_u8: null argument / no channels");
    int ps, nh, nw;
    if (int rc = innfer_chop_plan(H, W, patch, step, &ps, &nh, &nw, nullptr, nullptr)) return rc;
    const int step_int = (int)(ps * step);
    if (tile_begin < 0 || tile_count < 0 || tile_begin + tile_count > nh * nw)
        return set_error(INNFER_ERR_INVALID, "extract_tiles_u8: tile range [%d,+%d) outside %d tiles", tile_begin, tile_count, nh * nw);
    const long total = (long)tile_count * C * ps * ps;
    if (total == 0) return INNFER_OK;
    hipStream_t s = (hipStream_t)stream;
    if (tile_dtype == INNFER_F16)
        hipLaunchKernelGGL(k_extract_u8<f16>, dim3(blocks(total, 256)), dim3(256), 0, s, d_img, (f16*)d_tiles, C, H, W, ps, step_int, nw, tile_begin, total, normalize);
    else if (tile_dtype == INNFER_F32)
        hipLaunchKernelGGL(k_extract_u8<float>, dim3(blocks(total, 256)), dim3(256), 0, s, d_img, (float*)d_tiles, C, H, W, ps, step_int, nw, tile_begin, total, normalize);
    else return set_error(INNFER_ERR_INVALID, "extract_tiles_u8: bad dtype %d", tile_dtype);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

extern "C" int innfer_recompose_u8(const void* d_tiles, int dtype, int n, int C, int P, int height, int width, double step, int scale,
                                   int via_dtype, int denormalize, uint8_t* d_img, void* stream) {
    if (!d_tiles || !d_img) return set_error(INNFER_ERR_INVALID, "recompose_u8: null argument");
    if (step < 0.5 || step > 1.0) return set_error(INNFER_ERR_INVALID, "recompose_u8: step must be in [0.5,1]");
    if (n <= 0 || C <= 0 || C > 4 || P <= 0 || scale <= 0) return set_error(INNFER_ERR_INVALID, "recompose_u8: bad sizes");
    const int FH = scale * height, FW = scale * width;
    if (FH < P || FW < P) return set_error(INNFER_ERR_INVALID, "recompose_u8: patch %d larger than output %dx%d", P, FH, FW);
    const int ov = blend_overlap(P, step, scale);
    if (P - 2 * ov < 0) return set_error(INNFER_ERR_INVALID, "recompose_u8: overlap %d exceeds half of patch %d (reference raises too)", ov, P);
    const int eff = (int)(step * P), step_int = (int)(P * step);
    const int nh = 1 + (FH - P) / step_int + ((FH - P) % step_int != 0);
    const int nw = 1 + (FW - P) / step_int + ((FW - P) % step_int != 0);
    if (n != nh * nw) return set_error(INNFER_ERR_INVALID, "recompose_u8: one image of %dx%d tiles expected, got %d tiles", nh, nw, n);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((FW + 255) / 256, FH, 1), block(256);
#define RCU(TI, TO) hipLaunchKernelGGL((k_recompose<TI, TO, true>), grid, block, 0, s, (const TI*)d_tiles, (TO*)nullptr, C, P, FH, FW, eff, nh, nw, ov, d_img, denormalize)
    if (dtype == INNFER_F16 && via_dtype == INNFER_F16) RCU(f16, f16);
    else if (dtype == INNFER_F16 && via_dtype == INNFER_F32) RCU(f16, float);
    else if (dtype == INNFER_F32 && via_dtype == INNFER_F32) RCU(float, float);
    else if (dtype == INNFER_F32 && via_dtype == INNFER_F16) RCU(float, f16);
    else return set_error(INNFER_ERR_INVALID, "recompose_u8: bad dtype");
#undef RCU
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

extern "C" int innfer_u8hwc_to_nchw(const uint8_t* d_img, int H, int W, int C, int normalize,
                                    void* d_out, int out_dtype, void* stream) {
    const long hw = (long)H * W;
    if (hw <= 0 || C <= 0) return set_error(INNFER_ERR_INVALID, "u8hwc_to_nchw: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == INNFER_F16) hipLaunchKernelGGL((k_u8_to_nchw<uint8_t, f16>), dim3(blocks(hw, 256)), dim3(256), 0, s, d_img, (f16*)d_out, hw, C, normalize, 255.0f, 1);
    else if (out_dtype == INNFER_F32) hipLaunchKernelGGL((k_u8_to_nchw<uint8_t, float>), dim3(blocks(hw, 256)), dim3(256), 0, s, d_img, (float*)d_out, hw, C, normalize, 255.0f, 1);
    else return set_error(INNFER_ERR_INVALID, "u8hwc_to_nchw: bad dtype");
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

extern "C" int innfer_inthwc_to_nchw(const void* d_img, int bits, int H, int W, int C, int bgr2rgb, int normalize, float maxval,
                                     void* d_out, int out_dtype, void* stream) {
    const long hw = (long)H * W;
    if (!d_img || !d_out || hw <= 0 || C <= 0 || !(maxval > 0.f)) return set_error(INNFER_ERR_INVALID, "inthwc_to_nchw: bad arguments");
    if ((bits != 8 && bits != 16) || (out_dtype != INNFER_F16 && out_dtype != INNFER_F32))
        return set_error(INNFER_ERR_INVALID, "inthwc_to_nchw: bits %d (8, 16), dtype %d", bits, out_dtype);
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(blocks(hw, 256)), b(256);
    const int f = bgr2rgb ? 1 : 0;
    if (bits == 8 && out_dtype == INNFER_F16) hipLaunchKernelGGL((k_u8_to_nchw<uint8_t, f16>), g, b, 0, s, (const uint8_t*)d_img, (f16*)d_out, hw, C, normalize, maxval, f);
    else if (bits == 8) hipLaunchKernelGGL((k_u8_to_nchw<uint8_t, float>), g, b, 0, s, (const uint8_t*)d_img, (float*)d_out, hw, C, normalize, maxval, f);
    else if (out_dtype == INNFER_F16) hipLaunchKernelGGL((k_u8_to_nchw<uint16_t, f16>), g, b, 0, s, (const uint16_t*)d_img, (f16*)d_out, hw, C, normalize, maxval, f);
    else hipLaunchKernelGGL((k_u8_to_nchw<uint16_t, float>), g, b, 0, s, (const uint16_t*)d_img, (float*)d_out, hw, C, normalize, maxval, f);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

extern "C" int innfer_nchw_to_inthwc(const void* d_in, int in_dtype, int H, int W, int C, int rgb2bgr, int denormalize, int bits,
                                     void* d_img, void* stream) {
    const long hw = (long)H * W;
    if (!d_in || !d_img || hw <= 0 || C <= 0) return set_error(INNFER_ERR_INVALID, "nchw_to_inthwc: bad arguments");
    if ((bits != 8 && bits != 16) || (in_dtype != INNFER_F16 && in_dtype != INNFER_F32))
        return set_error(INNFER_ERR_INVALID, "nchw_to_inthwc: bits %d (8, 16), dtype %d", bits, in_dtype);
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(blocks(hw, 256)), b(256);
    const int f = rgb2bgr ? 1 : 0;
    if (bits == 8 && in_dtype == INNFER_F16) hipLaunchKernelGGL((k_nchw_to_u8<f16, uint8_t>), g, b, 0, s, (const f16*)d_in, (uint8_t*)d_img, hw, C, denormalize, 255.0f, f);
    else if (bits == 8) hipLaunchKernelGGL((k_nchw_to_u8<float, uint8_t>), g, b, 0, s, (const float*)d_in, (uint8_t*)d_img, hw, C, denormalize, 255.0f, f);
    else if (in_dtype == INNFER_F16) hipLaunchKernelGGL((k_nchw_to_u8<f16, uint16_t>), g, b, 0, s, (const f16*)d_in, (uint16_t*)d_img, hw, C, denormalize, 65535.0f, f);
    else hipLaunchKernelGGL((k_nchw_to_u8<float, uint16_t>), g, b, 0, s, (const float*)d_in, (uint16_t*)d_img, hw, C, denormalize, 65535.0f, f);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

extern "C" int innfer_nchw_to_u8hwc(const void* d_in, int in_dtype, int H, int W, int C, int denormalize,
                                    uint8_t* d_img, void* stream) {
    const long hw = (long)H * W;
    if (hw <= 0 || C <= 0) return set_error(INNFER_ERR_INVALID, "nchw_to_u8hwc: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == INNFER_F16) hipLaunchKernelGGL((k_nchw_to_u8<f16, uint8_t>), dim3(blocks(hw, 256)), dim3(256), 0, s, (const f16*)d_in, d_img, hw, C, denormalize, 255.0f, 1);
    else if (in_dtype == INNFER_F32) hipLaunchKernelGGL((k_nchw_to_u8<float, uint8_t>), dim3(blocks(hw, 256)), dim3(256), 0, s, (const float*)d_in, d_img, hw, C, denormalize, 255.0f, 1);
    else return set_error(INNFER_ERR_INVALID, "nchw_to_u8hwc: bad dtype");
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

extern "C" int innfer_srgb_to_linear(const uint8_t* d_in, float* d_out, size_t n, void* stream) {
    if (!d_in || !d_out) return set_error(INNFER_ERR_INVALID, "srgb_to_linear: null argument");
    if (n == 0) return INNFER_OK;
    hipLaunchKernelGGL(k_srgb2linear, dim3(blocks((long)n, 256)), dim3(256), 0, (hipStream_t)stream, d_in, d_out, (long)n);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

extern "C" int innfer_linear_to_srgb(const float* d_in, uint8_t* d_out, size_t n, void* stream) {
    if (!d_in || !d_out) return set_error(INNFER_ERR_INVALID, "linear_to_srgb: null argument");
    if (n == 0) return INNFER_OK;
    hipLaunchKernelGGL(k_linear2srgb, dim3(blocks((long)n, 256)), dim3(256), 0, (hipStream_t)stream, d_in, d_out, (long)n);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}
