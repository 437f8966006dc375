// Diagnostic driver of csrc/pan_scpa.hip (never shipped): one SCPA block on random data of a 540 x 960 frame (or H W N from the command line), both forms of the block kernel
// -- one 8-wave workgroup per CU on 16 x 32 tiles, two 4-wave workgroups per CU on 8 x 32 tiles --, timed with HIP events, and the in-kernel phase stamps of the
// INNFER_STAMPS build: wave 0's shader-clock cycles per tile in P1 / P2a / P2b / P3 (each up to and including the barrier that ends it), averaged over the workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DINNFER_STAMPS -Wno-unused-function -Iinclude -o scripts/micro/scpa_micro scripts/r6/scpa_micro.cpp && scripts/micro/scpa_micro
#include "../../innfer_amd/csrc/pan_scpa.hip"
#include <random>
#include <vector>
namespace innfer {
int set_error(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); return code; }
bool gt_on() { return false; }
void gt_begin(hipStream_t) {}
void gt_end(hipStream_t, const char*, double, double) {}
}
using namespace innfer;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int H = argc > 2 ? atoi(argv[1]) : 540, W = argc > 2 ? atoi(argv[2]) : 960, N = argc > 3 ? atoi(argv[3]) : 1, reps = 20;
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> d(-1.f, 1.f);
    auto rnd = [&](size_t n, float sc) { std::vector<float> v(n); for (auto& x : v) x = sc * d(rng); return v; };
    const auto c1a = rnd(800, 0.15f), c1b = rnd(800, 0.15f), k1 = rnd(3600, 0.07f), k2 = rnd(400, 0.2f), k2b = rnd(20, 0.1f), k3 = rnd(3600, 0.07f), k4 = rnd(3600, 0.07f), c3 = rnd(1600, 0.15f);
    std::vector<char> blob(pan_scpa_blob_bytes());
    pan_scpa_pack(c1a.data(), c1b.data(), k1.data(), k2.data(), k2b.data(), k3.data(), k4.data(), c3.data(), blob.data());
    const long px = (long)N * H * W, G = px * 32;
    std::vector<f16> x((size_t)2 * G);
    for (long i = 0; i < 2 * G; ++i) x[i] = ((i / 32) % 2 == 0 || i < G || (i % 32) < 8) ? (f16)d(rng) : (f16)0.f;          // (group 1: channels 32..39 real, the rest zero)
    for (long i = G; i < 2 * G; ++i) if ((i % 32) >= 8) x[i] = (f16)0.f;
    void *d_in, *d_out, *d_blob;
    unsigned long long* d_st;
    CK(hipMalloc(&d_in, (size_t)4 * G)); CK(hipMalloc(&d_out, (size_t)4 * G)); CK(hipMalloc(&d_blob, blob.size())); CK(hipMalloc((void**)&d_st, 1024 * 7 * 8));
    CK(hipMemcpy(d_in, x.data(), (size_t)4 * G, hipMemcpyHostToDevice)); CK(hipMemcpy(d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
    g_scpa_stamps = d_st;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int duo = 0; duo < 2; ++duo) {
        for (int i = 0; i < 3; ++i) if (pan_scpa_launch((const f16*)d_in, (f16*)d_out, G, d_blob, N, H, W, 0, 0, 0, duo)) return 1;
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) if (pan_scpa_launch((const f16*)d_in, (f16*)d_out, G, d_blob, N, H, W, 0, 0, 0, duo)) return 1;
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemset(d_st, 0, 1024 * 7 * 8));
        if (pan_scpa_launch((const f16*)d_in, (f16*)d_out, G, d_blob, N, H, W, 0, 0, 0, duo)) return 1;
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> st(1024 * 7);
        CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        double acc[4] = {0, 0, 0, 0}, tiles = 0;
        int wgs = 0;
        double clk = 0;
        for (int b = 0; b < 1024; ++b) if (st[b * 7 + 4]) { for (int i = 0; i < 4; ++i) acc[i] += (double)st[b * 7 + i]; tiles += (double)st[b * 7 + 4]; clk += 100.0 * (double)st[b * 7 + 5] / (double)st[b * 7 + 6]; ++wgs; }
        const double tot = acc[0] + acc[1] + acc[2] + acc[3];
        printf("%s  %dx%dx%d  %.2f us per launch;  %d workgroups, %.2f tiles each;  cycles per tile (wave 0, barrier included): P1 %.0f  P2a %.0f  P2b %.0f  P3 %.0f  = %.0f;  clock held %.0f MHz\n",
               duo ? "two 4-wave workgroups per CU, 8 x 32 tiles " : "one 8-wave workgroup per CU, 16 x 32 tiles", N, H, W, 1e3 * ms / reps, wgs, tiles / wgs,
               acc[0] / tiles, acc[1] / tiles, acc[2] / tiles, acc[3] / tiles, tot / tiles, clk / wgs);
    }
    return 0;
}
