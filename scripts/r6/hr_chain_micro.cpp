// Diagnostic driver of csrc/hr_chain.hip (never shipped): the chained kernel alone on random buffers of a 1080p -> 4K frame's last stage (h x w = 2160 x 3840 ->
// 4320 x 7680), timed with HIP events, with the kernel's ablation bits (INNFER_ABL, see ChainP.abl) -- results are meaningless, only the times mean anything.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DINNFER_ABLATE -o scripts/micro/hr_chain_micro scripts/r6/hr_chain_micro.cpp && for a in 0 1 2 4 8 ...; do INNFER_ABL=$a scripts/micro/hr_chain_micro; done
#include "../../innfer_amd/csrc/hr_chain.hip"
#include <random>
namespace innfer {
int set_error(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); return code; }
bool gt_on() { return false; }
void gt_begin(hipStream_t) {}
void gt_end(hipStream_t, const char*, double, double) {}
bool conv_fuse_last_ok(const ConvLaunch&) { return true; }
}
using namespace innfer;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int H = argc > 2 ? atoi(argv[1]) : 4320, W = argc > 2 ? atoi(argv[2]) : 7680, reps = argc > 3 ? atoi(argv[3]) : 20;
    const int h = H / 2, w = W / 2;
    std::mt19937 rng(1);
    auto fill16 = [&](size_t n, float sc) { std::vector<f16> v(n); std::uniform_real_distribution<float> d(-sc, sc); for (auto& x : v) x = (f16)d(rng); return v; };
    auto up = [&](const void* src, size_t bytes, void** dst) { if (hipMalloc(dst, bytes) != hipSuccess) return 1; return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1; };
    void *d_in, *d_wup, *d_bup, *d_whr, *d_bhr, *d_flw, *d_flb, *d_side, *d_out;
    { auto v = fill16((size_t)2 * h * w * 32, 1.f); if (up(v.data(), v.size() * 2, &d_in)) return 1; }
    { auto v = fill16(65536, 0.06f); if (up(v.data(), v.size() * 2, &d_wup)) return 1; }
    { auto v = fill16(2 * 9 * 64 * 32, 0.04f); if (up(v.data(), v.size() * 2, &d_whr)) return 1; }
    { auto v = fill16(2048, 0.04f); if (up(v.data(), v.size() * 2, &d_flw)) return 1; }
    std::vector<float> b(256, 0.01f);
    if (up(b.data(), 1024, &d_bup) || up(b.data(), 256, &d_bhr) || up(b.data(), 12, &d_flb)) return 1;
    CK(hipMalloc(&d_side, (size_t)(H / 16) * (W / 32) * 192 * 3 * 4));
    CK(hipMalloc(&d_out, (size_t)3 * H * W * 2));
    ConvLaunch L{};
    L.wpk = (const f16*)d_whr; L.bias = (const float*)d_bhr; L.K = 64; L.C = 64; L.N = 1; L.H = H; L.W = W; L.act = 1; L.rowp = 1; L.out_mode = OUT_SLAB; L.y1 = H;
    L.fuse_w = (const f16*)d_flw; L.fuse_bias = (const float*)d_flb; L.fuse_side = (float*)d_side; L.fuse_out = d_out; L.fuse_oc = 3; L.fuse_out_mode = 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) if (hr_chain_launch(L, (const f16*)d_in, (long)h * w * 32, (const f16*)d_wup, (const float*)d_bup, 1, 0)) return 1;
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) { L.rev = i & 1; if (hr_chain_launch(L, (const f16*)d_in, (long)h * w * 32, (const f16*)d_wup, (const float*)d_bup, 1, 0)) return 1; }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("abl=%s  %dx%d  %.4f ms per launch (kernel + rim pass)\n", getenv("INNFER_ABL") ? getenv("INNFER_ABL") : "0", H, W, ms / reps);
    return 0;
}
