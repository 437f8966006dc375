// `-cf` colour fix on the GPU: replaces color_fix (utils/utils.py:278-315) with its srgb2linear /
// linear2srgb (utils/colors.py:29-60) ends.  The reference calls OpenCV twice (cv2.resize INTER_CUBIC and
// cv2.GaussianBlur 3x3, sigma 0); OpenCV is neither vendored nor installed here, so these kernels follow
// the published float32 algorithms (oracle/colorfix.py states them) and parity with OpenCV itself is
// UNPINNED.  All images are HWC, the two uint8 inputs and the uint8 output live on the device.
//   1. lin_b = srgb2linear(B) (kept: it is needed again at the end); diff = srgb2linear(A) - cubic(lin_b -> size A)
//   2. blur  = gauss3(diff)            rows, then columns, BORDER_REFLECT_101
//   3. out   = linear2srgb(cubic(blur -> size B) + lin_b)
// HBM-bound pointwise / gather work: one thread per output element, contraction off so that the float32
// operation order is the one written down (and mirrored by the oracle).
#include "common.h"

#pragma clang fp contract(off)

using namespace innfer;

namespace {

// The two transfer functions as written down (colors.py:29-60 on uint8 / on the float result)...
__device__ __forceinline__ float srgb2lin_eval(uint8_t v) {
    const float l = __fdiv_rn((float)v, 255.0f);
    return l <= 0.04045f ? __fdiv_rn(l, 12.92f) : powf(__fdiv_rn(__fadd_rn(l, 0.055f), 1.055f), 2.4f);
}

__device__ __forceinline__ uint8_t lin2srgb_eval(float x) {
    float s = fminf(fmaxf(x, 0.0f), 1.0f);
    s = s <= 0.0031308f ? __fmul_rn(s, 12.92f) : __fsub_rn(__fmul_rn(1.055f, powf(s, (float)(1.0 / 2.4))), 0.055f);
    s = fminf(fmaxf(__fmul_rn(s, 255.0f), 0.0f), 255.0f);
    return (uint8_t)(int)s;
}

// ... and as the image kernels apply them: both have only 256 outcomes, so the powf leaves the per-pixel path (it was what color_fix's time went to).
// g_s2l[v] = srgb2lin_eval(v); g_thr[k] = the smallest float in [0, 1] that lin2srgb_eval maps to a code >= k (k = 1 .. 255; found by bisection over
// the float bit patterns with lin2srgb_eval itself, once per device) -- linear2srgb(x) is then the number of thresholds <= clamp(x), eight compares.
__device__ float g_s2l[256];
__device__ float g_thr[256];

__global__ void k_build_tables() {
    const int k = threadIdx.x;
    g_s2l[k] = srgb2lin_eval((uint8_t)k);
    unsigned lo = 0u, hi = 0x3F800000u;                 // bit patterns of 0.0f and 1.0f: non-negative floats order like their bits
    if (k == 0) { g_thr[0] = 0.0f; return; }
    if (lin2srgb_eval(__uint_as_float(hi)) < k) { g_thr[k] = 2.0f; return; }       // (code 255 is reached at 1.0: never taken)
    while (lo < hi) {                                   // invariant: eval(hi) >= k
        const unsigned mid = lo + ((hi - lo) >> 1);
        if (lin2srgb_eval(__uint_as_float(mid)) >= k) hi = mid; else lo = mid + 1;
    }
    g_thr[k] = __uint_as_float(hi);
}

__device__ __forceinline__ float srgb2lin(uint8_t v) { return g_s2l[v]; }

// thr: the 256 thresholds in LDS
__device__ __forceinline__ uint8_t lin2srgb(float x, const float* thr) {
    const float s = fminf(fmaxf(x, 0.0f), 1.0f);
    int code = 0;
#pragma unroll
    for (int step = 128; step >= 1; step >>= 1) code += (thr[code + step] <= s) ? step : 0;
    return (uint8_t)code;
}

// OpenCV's cubic taps for destination index d: first source index (s - 1) and the four weights
__device__ __forceinline__ void cubic_taps(int d, float scale, int& s, float w[4]) {
    const float A = -0.75f;
    float f = ((float)d + 0.5f) * scale - 0.5f;
    const float fl = floorf(f);
    s = (int)fl;
    const float t = f - fl;
    w[0] = ((A * (t + 1.f) - 5.f * A) * (t + 1.f) + 8.f * A) * (t + 1.f) - 4.f * A;
    w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    w[2] = ((A + 2.f) * (1.f - t) - (A + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
    w[3] = 1.f - w[0] - w[1] - w[2];
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// horizontal pass, then vertical pass, as cv2.resize does for float images
// (SRC = uint8_t: the source is an sRGB image, linearised through the table as it is read)
__device__ __forceinline__ float cs_load(const float* p, long o) { return p[o]; }
__device__ __forceinline__ float cs_load(const uint8_t* p, long o) { return srgb2lin(p[o]); }
template <typename SRC>
__device__ __forceinline__ float cubic_sample(const SRC* src, int Hs, int Ws, int C, int c, int y, int x, float sy, float sx) {
    int x0, y0;
    float wx[4], wy[4];
    cubic_taps(x, sx, x0, wx);
    cubic_taps(y, sy, y0, wy);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const SRC* row = src + (long)clampi(y0 - 1 + j, 0, Hs - 1) * Ws * C + c;
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) r = r + cs_load(row, (long)clampi(x0 - 1 + i, 0, Ws - 1) * C) * wx[i];
        acc = acc + r * wy[j];
    }
    return acc;
}

__global__ void k_lin(const uint8_t* in, float* out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = srgb2lin(in[i]);
}

// diff = srgb2linear(A) - (scaling ? cubic(srgb2linear(B) -> A's size) : srgb2linear(B)); B is linearised through the table as it is read
__global__ void k_diff(const uint8_t* a, const uint8_t* b8, int hA, int wA, int hB, int wB, int C, int scaling, float* diff) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)hA * wA * C) return;
    const int c = (int)(i % C), x = (int)((i / C) % wA), y = (int)(i / ((long)C * wA));
    const float b = scaling ? cubic_sample(b8, hB, wB, C, c, y, x, (float)hB / (float)hA, (float)wB / (float)wA) : srgb2lin(b8[i]);
    diff[i] = srgb2lin(a[i]) - b;
}

// one pass of the [0.25, 0.5, 0.25] filter along x (dir 0) or y (dir 1), BORDER_REFLECT_101
__global__ void k_gauss3(const float* in, int H, int W, int C, int dir, float* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)H * W * C) return;
    const int c = (int)(i % C), x = (int)((i / C) % W), y = (int)(i / ((long)C * W));
    const int n = dir ? H : W, p = dir ? y : x;
    int lo = p - 1, hi = p + 1;
    if (lo < 0) lo = n > 1 ? 1 : 0;
    if (hi >= n) hi = n > 1 ? n - 2 : 0;
    const long step = dir ? (long)W * C : C;
    const float* base = in + i - (long)p * step;
    (void)c;
    out[i] = in[i] * 0.5f + (base[(long)lo * step] + base[(long)hi * step]) * 0.25f;
}

// out = linear2srgb((scaling ? cubic(blur -> B's size) : blur) + srgb2linear(B))
__global__ void k_finish(const float* blur, const uint8_t* b8, int hA, int wA, int hB, int wB, int C, int scaling, uint8_t* out) {
    __shared__ float thr[256];
    thr[threadIdx.x] = g_thr[threadIdx.x];
    __syncthreads();
    // one thread per pixel: the cubic taps and weights are the same for its C channels (same arithmetic per channel as cubic_sample)
    const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (long)hB * wB) return;
    const int x = (int)(pix % wB), y = (int)(pix / wB);
    if (!scaling) {
        for (int c = 0; c < C; ++c) out[pix * C + c] = lin2srgb(blur[pix * C + c] + srgb2lin(b8[pix * C + c]), thr);
        return;
    }
    int x0, y0;
    float wx[4], wy[4];
    cubic_taps(x, (float)wA / (float)wB, x0, wx);
    cubic_taps(y, (float)hA / (float)hB, y0, wy);
    long ro[4]; int co[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { ro[j] = (long)clampi(y0 - 1 + j, 0, hA - 1) * wA * C; co[j] = clampi(x0 - 1 + j, 0, wA - 1) * C; }
    for (int c = 0; c < C; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float r = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) r = r + blur[ro[j] + co[i] + c] * wx[i];
            acc = acc + r * wy[j];
        }
        out[pix * C + c] = lin2srgb(acc + srgb2lin(b8[pix * C + c]), thr);
    }
}

// linear_resize (utils.py:267-276): out = linear2srgb(cubic(srgb2linear(img) -> (oh, ow)))
__global__ void k_resize_finish(const float* lin, int h, int w, int C, int oh, int ow, uint8_t* out) {
    __shared__ float thr[256];
    thr[threadIdx.x] = g_thr[threadIdx.x];
    __syncthreads();
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)oh * ow * C) return;
    const int c = (int)(i % C), x = (int)((i / C) % ow), y = (int)(i / ((long)C * ow));
    out[i] = lin2srgb(cubic_sample(lin, h, w, C, c, y, x, (float)h / (float)oh, (float)w / (float)ow), thr);
}

// the tables are built once per device (the first call waits for them: later calls may come on other streams)
int ensure_tables(hipStream_t s) {
    static bool built[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    if (!built[dev & 63]) {
        hipLaunchKernelGGL(k_build_tables, dim3(1), dim3(256), 0, s);
        INNFER_HIP(hipGetLastError());
        INNFER_HIP(hipStreamSynchronize(s));
        built[dev & 63] = true;
    }
    return INNFER_OK;
}

inline unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }
inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t innfer_color_fix_workspace_bytes(int hA, int wA, int hB, int wB, int C) {
    if (hA <= 0 || wA <= 0 || hB <= 0 || wB <= 0 || C <= 0) return 0;
    (void)hB; (void)wB;                                 // two fp32 planes of A's size (difference, blur); nothing of B's size is kept
    return 2 * al((size_t)hA * wA * C * 4);
}

extern "C" int innfer_linear_resize(const uint8_t* d_img, int h, int w, int C, uint8_t* d_out, int oh, int ow, void* d_ws, size_t ws_bytes, void* stream) {
    if (!d_img || !d_out || !d_ws || h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || C <= 0 || C > 4) return set_error(INNFER_ERR_INVALID, "linear_resize: bad arguments");
    if (ws_bytes < (size_t)h * w * C * 4) return set_error(INNFER_ERR_WORKSPACE, "linear_resize: workspace %zu < %zu bytes", ws_bytes, (size_t)h * w * C * 4);
    hipStream_t s = (hipStream_t)stream;
    if (int rc = ensure_tables(s)) return rc;
    const long n = (long)h * w * C, no = (long)oh * ow * C;
    hipLaunchKernelGGL(k_lin, dim3(nblk(n)), dim3(256), 0, s, d_img, (float*)d_ws, n);
    hipLaunchKernelGGL(k_resize_finish, dim3(nblk(no)), dim3(256), 0, s, (const float*)d_ws, h, w, C, oh, ow, d_out);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

extern "C" int innfer_color_fix(const uint8_t* d_a, int hA, int wA, const uint8_t* d_b, int hB, int wB, int C, uint8_t* d_out,
                                void* d_ws, size_t ws_bytes, void* stream) {
    if (!d_a || !d_b || !d_out || !d_ws) return set_error(INNFER_ERR_INVALID, "color_fix: null argument");
    if (hA <= 0 || wA <= 0 || hB <= 0 || wB <= 0 || C <= 0 || C > 4) return set_error(INNFER_ERR_INVALID, "color_fix: bad shape");
    const int scaling = hA < hB && wA < wB;
    if (!scaling && (hA != hB || wA != wB))
        return set_error(INNFER_ERR_INVALID, "color_fix: images of %dx%d and %dx%d cannot be subtracted (the reference raises too)", hA, wA, hB, wB);
    if (ws_bytes < innfer_color_fix_workspace_bytes(hA, wA, hB, wB, C))
        return set_error(INNFER_ERR_WORKSPACE, "color_fix: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    if (int rc = ensure_tables(s)) return rc;
    const long nA = (long)hA * wA * C, nB = (long)hB * wB * C;
    float* t0 = (float*)d_ws;                          // (the linear copy of B the first version kept is gone: B is linearised through the table where it is read)
    float* t1 = (float*)((char*)t0 + al((size_t)nA * 4));
    (void)nB;
    hipLaunchKernelGGL(k_diff, dim3(nblk(nA)), dim3(256), 0, s, d_a, d_b, hA, wA, hB, wB, C, scaling, t0);
    hipLaunchKernelGGL(k_gauss3, dim3(nblk(nA)), dim3(256), 0, s, (const float*)t0, hA, wA, C, 0, t1);
    hipLaunchKernelGGL(k_gauss3, dim3(nblk(nA)), dim3(256), 0, s, (const float*)t1, hA, wA, C, 1, t0);
    hipLaunchKernelGGL(k_finish, dim3(nblk((long)hB * wB)), dim3(256), 0, s, (const float*)t0, d_b, hA, wA, hB, wB, C, scaling, d_out);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}
