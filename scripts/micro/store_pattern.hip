// Store-address pattern of the conv epilogue: a wave writes a 1-KiB block (16 pixels x 64 B) with global_store_dwordx4.
//   pattern 0: lane (lg = lane / 16, li = lane % 16) writes the 16-B quarter lg of pixel li  -> base + li * 64 + lg * 16
//              (what the MFMA accumulator layout gives: consecutive lanes are 64 B apart)
//   pattern 1: lane l writes piece l -> base + l * 16 (consecutive lanes, consecutive 16-B pieces)
//   pattern 2: pattern 1 after moving the data with 4 ds_bpermute_b32 (what the epilogue would have to do)
// hipcc --offload-arch=gfx950 -O3 -o scripts/micro/store_pattern scripts/micro/store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512) void k(u32x4* out, long blocks_per_wave, long nblocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long w = (long)blockIdx.x * 8 + wave;
    u32x4 v = {(unsigned)lane, (unsigned)w, 3u, 4u};
    for (long b = 0; b < blocks_per_wave; ++b) {
        const long blk = (w * blocks_per_wave + b) % nblocks;        // nblocks small: the target stays in L2, the store path itself is timed
        u32x4* base = out + blk * 64;
        v[2] += (unsigned)b;
        if (PAT == 0) {
            const int li = lane & 15, lg = lane >> 4;
            base[li * 4 + lg] = v;
        } else if (PAT == 1) {
            base[lane] = v;
        } else {
            const int src = ((lane & 3) * 16 + (lane >> 2)) * 4;
            u32x4 t;
            for (int d = 0; d < 4; ++d) t[d] = __builtin_amdgcn_ds_bpermute(src, v[d]);
            base[lane] = t;
        }
    }
}

int main() {
    const long total = 2L << 30;
    u32x4* d;
    hipMalloc(&d, total);
    for (long bytes : {total, 8L << 20}) {
    const long nblocks = bytes / 1024;
    printf("target %ld MB\n", bytes >> 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bpw : {6, 48}) {
        const long waves = (total / 1024 + bpw - 1) / bpw, grid = (waves + 7) / 8;
        for (int pat = 0; pat < 3; ++pat) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, d, (long)bpw, nblocks);
                if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, d, (long)bpw, nblocks);
                if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, d, (long)bpw, nblocks);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("blocks/wave %2d pattern %d: %7.3f ms  %6.2f TB/s\n", bpw, pat, best, total / best / 1e9);
        }
    }
    }
    return 0;
}
