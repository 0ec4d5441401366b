// Microbenchmark: can one wave's fp32 MFMAs and another wave's VALU work share a SIMD?
// Workgroup = 8 waves: waves 0-3 land on SIMD 0-3, waves 4-7 on the same SIMDs.
// mode bit 0: waves 0-3 run a chain of v_mfma_f32_16x16x4_f32; bit 1: waves 4-7 run VALU FMAs;
// bit 2: waves 4-7 run MFMAs too (two MFMA waves per SIMD).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) k(int mode, int iters, float *out)
{
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if ((wave < 4 && (mode & 1)) || (wave >= 4 && (mode & 4))) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
        float x = threadIdx.x * 1e-3f, y = 1.0001f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
            }
        }
        r = a0[0] + a1[1];
    } else if (wave >= 4 && (mode & 2)) {
        float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 32; ++j) { // 128 independent-ish FMAs per iteration
                v0 = __builtin_fmaf(v0, 1.0001f, 0.5f);
                v1 = __builtin_fmaf(v1, 1.0002f, 0.25f);
                v2 = __builtin_fmaf(v2, 0.9999f, 0.125f);
                v3 = __builtin_fmaf(v3, 0.9998f, 0.0625f);
            }
        }
        r = v0 + v1 + v2 + v3;
    }
    if (r == 12345.f) out[threadIdx.x] = r;
}

int main()
{
    float *out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode : {1, 2, 3, 4, 5, 1, 2, 3}) {
        k<<<256, 512>>>(mode, 100, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<256, 512>>>(mode, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d: %.3f ms  (MFMA waves: 16 x 32 cyc = 512 cyc/iter; VALU waves: 128 x 4 = 512 cyc/iter)\n", mode, ms);
    }
    return 0;
}
