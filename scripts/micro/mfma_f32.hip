// What does v_mfma_f32_16x16x4_f32 sustain in the shape of f32conv_tiled's step loop?  16 MFMAs (4 x 4 register blocking) per step on operands that are
// (REG) loop-invariant registers, (LDS) 8 ds_read_b32 per step one step ahead; 1 or 2 workgroups of 4 waves per CU (WPS waves per SIMD); accumulators are the
// compiler's choice (AGPRs at this pressure).  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f32 mfma_f32.hip ; run: ./mfma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int LDSR>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 0.001f * (float)((i * 2654435761u) >> 20);
    __syncthreads();
    f32x4 acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
    float fa[4], fb[4];
    for (int q = 0; q < 4; ++q) { fa[q] = lds[(li * 4 + lg) + 64 * q]; fb[q] = lds[4096 + q * 1024 + li + lg * 336]; }
    for (int it = 0; it < iters; ++it) {
        float na[4], nb[4];
        if constexpr (LDSR) {
            const int o = (it & 15) * 256;
#pragma unroll
            for (int q = 0; q < 4; ++q) { na[q] = lds[o + (li * 4 + lg) + 64 * q]; nb[q] = lds[4096 + q * 1024 + li + lg * 336 + (it & 7)]; }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
        if constexpr (LDSR) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { fa[q] = na[q]; fb[q] = nb[q]; }
        }
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][3];
    if (s == 12345.f) out[threadIdx.x] = s;
}

// (Q) the same 64 MFMAs on 8 ds_read_b128 per FOUR steps: a lane's four k-group values of one channel / pixel as one 16-byte read (slot-swizzled rows of 64 bytes)
__global__ __launch_bounds__(256) void kq(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 0.001f * (float)((i * 2654435761u) >> 20);
    __syncthreads();
    f32x4 acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
    f32x4 fa[4], fb[4];
    const int sw = (li * 4 + (lg ^ ((li >> 2) & 3))) * 4;
    for (int q = 0; q < 4; ++q) { fa[q] = *(const f32x4*)(lds + sw + 256 * q); fb[q] = *(const f32x4*)(lds + 4096 + q * 1024 + sw); }
    for (int it = 0; it < iters; it += 4) {
        f32x4 na[4], nb[4];
        const int o = (it & 15) * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) { na[q] = *(const f32x4*)(lds + o + sw + 256 * q); nb[q] = *(const f32x4*)(lds + 4096 + q * 1024 + sw + (it & 28) * 4); }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a][e], fb[b][e], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) { fa[q] = na[q]; fb[q] = nb[q]; }
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][3];
    if (s == 12345.f) out[threadIdx.x] = s;
}

void runq(int wgs_per_cu) {
    float* out; hipMalloc(&out, 4096);
    const int iters = 20000, cus = 256;
    const size_t lds = wgs_per_cu == 1 ? 100 * 1024 : (wgs_per_cu == 2 ? 66 * 1024 : 40 * 1024);
    hipFuncSetAttribute((const void*)kq, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kq<<<cus * wgs_per_cu, 256, lds>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kq<<<cus * wgs_per_cu, 256, lds>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)cus * wgs_per_cu * 4 * iters * 16 * 2048.0;
    printf("%-44s %d wave(s)/SIMD  %8.3f ms  %7.1f TFLOP/s\n", "64 MFMA / 4 steps, 8 ds_read_b128", wgs_per_cu, ms, fl / ms / 1e9);
    hipFree(out);
}

template <int LDSR>
void run(int wgs_per_cu, const char* what) {
    float* out; hipMalloc(&out, 4096);
    const int iters = 20000, cus = 256;
    const size_t lds = wgs_per_cu == 1 ? 100 * 1024 : (wgs_per_cu == 2 ? 66 * 1024 : 40 * 1024);        // the LDS size sets the co-residency
    hipFuncSetAttribute((const void*)k<LDSR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<LDSR><<<cus * wgs_per_cu, 256, lds>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<LDSR><<<cus * wgs_per_cu, 256, lds>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)cus * wgs_per_cu * 4 * iters * 16 * 2048.0;
    printf("%-44s %d wave(s)/SIMD  %8.3f ms  %7.1f TFLOP/s\n", what, wgs_per_cu, ms, fl / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 3; ++w) run<0>(w, "16 MFMA / step, register operands");
    for (int w = 1; w <= 3; ++w) run<1>(w, "16 MFMA / step, 8 ds_read_b32 / step");
    for (int w = 1; w <= 3; ++w) runq(w);
    return 0;
}
