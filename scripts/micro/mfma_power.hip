// Sustained MFMA rate under the package power cap: 16x16x32 f16 vs 32x32x16 f16, registers only, and with
// the conv kernel's ds_read_b128 : MFMA ratio.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int LDSR, int RND>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 65536 / 2; i += 512) {                                       // RND: activations-like random fp16, else a few constants
        unsigned h = (i + 1) * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        ((_Float16*)lds)[i] = RND ? (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.f)) : (_Float16)(0.001f * (i & 7));
    }
    __syncthreads();
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = RND ? ((const _Float16*)lds)[(threadIdx.x * 8 + e) & 32767] : (_Float16)(0.01f * (lane + e));
        b[e] = RND ? ((const _Float16*)lds)[(threadIdx.x * 8 + e + 4096) & 32767] : (_Float16)(0.02f * (lane - e));
    }
    const char* base = lds + ((threadIdx.x * 16) & 32767);
    if constexpr (MODE == 0) {
        f32x4 acc[16];
        for (int t = 0; t < 16; ++t) acc[t] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
            f16x8 av[4] = {a, a, a, a}, bv[4] = {b, b, b, b};
            if constexpr (LDSR) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { av[q] = *(const f16x8*)(base + q * 1024 + (it & 7) * 4096); bv[q] = *(const f16x8*)(base + 32768 + q * 1024 - (it & 7) * 1024 + 7168); }
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[t & 3], bv[t >> 2], acc[t], 0, 0, 0);
        }
        float s = 0;
        for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][3];
        if (s == 12345.f) out[threadIdx.x] = s;
    } else {
        f32x16 acc[4];
        for (int t = 0; t < 4; ++t) for (int e = 0; e < 16; ++e) acc[t][e] = 0;
        for (int it = 0; it < iters; ++it) {
            f16x8 av[4] = {a, a, a, a}, bv[4] = {b, b, b, b};
            if constexpr (LDSR) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { av[q] = *(const f16x8*)(base + q * 1024 + (it & 7) * 4096); bv[q] = *(const f16x8*)(base + 32768 + q * 1024 - (it & 7) * 1024 + 7168); }
            }
            // same flops per iteration as MODE 0: 2x2 tiles of 32x32, two 16-deep k steps = 8 MFMAs of 32768 flops
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[(t & 1) + 2 * (t >> 2)], bv[((t >> 1) & 1) + 2 * (t >> 2)], acc[t & 3], 0, 0, 0);
        }
        float s = 0;
        for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][15];
        if (s == 12345.f) out[threadIdx.x] = s;
    }
}

template <int MODE, int LDSR, int RND>
void run(const char* name, float* out, double secs) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, LDSR, RND>), dim3(256), dim3(512), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, LDSR, RND>), dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms1; hipEventElapsedTime(&ms1, e0, e1);
    int reps = (int)(secs * 1000.0 / ms1) + 1;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<MODE, LDSR, RND>), dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1);
    // sample the clock / power while the queue drains
    FILE* f = popen("for i in 1 2 3; do sleep 1; rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Package Power' | sed 's/.*: //' | tr '\\n' ' '; done", "r");
    char buf[512] = {0}; if (f) { fread(buf, 1, 511, f); pclose(f); }
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 8 * iters * 16 * 16384.0 * reps;
    printf("%-28s %8.1f TFLOP/s  cold single launch %8.1f TFLOP/s   [%s]\n", name, flops / ms / 1e9, 256.0 * 8 * iters * 16 * 16384.0 / ms1 / 1e9, buf);
    fflush(stdout);
}
int main() {
    float* out; hipMalloc(&out, 4096);
    run<0, 0, 0>("16x16x32 regs const", out, 4);
    run<0, 0, 1>("16x16x32 regs random", out, 4);
    run<1, 0, 1>("32x32x16 regs random", out, 4);
    run<0, 1, 1>("16x16x32 lds random", out, 4);
    run<1, 1, 1>("32x32x16 lds random", out, 4);
    run<0, 1, 0>("16x16x32 lds const", out, 4);
    return 0;
}
