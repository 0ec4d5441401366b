// Microbenchmark: v_mfma_f32_4x4x1_16b_f32 with the A-matrix broadcast (cbsz = 4, abid = b).
// (1) semantics: with lane l = 4*block + i holding A[i] of block `block` and B[j] of block `block`,
//     cbsz = 4 / abid = b makes every block use block b's A: D[vgpr i][lane l] += A_b[i] * B[l].
//     So one instruction is a 4-row x 64-column rank-1 update, and a 4-row group of a sparse conv
//     needs no padding to 16 rows.
// (2) rate: chains of dependent instructions with 1, 2, 4 independent accumulators per wave, 1 wave
//     per SIMD and 2 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 mfma4x4.hip -o mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// out[i][l] = sum_k a[i][k] * w[k][l], 4 rows x 64 k x 64 cols
__global__ void __launch_bounds__(64) sem(const float *a, const float *w, float *out)
{
    const int l = threadIdx.x, i = l & 3, b = l >> 2;
    // lane (i, b) holds a[i][4b .. 4b+3]
    float av[4];
    for (int j = 0; j < 4; ++j) av[j] = a[i * 64 + 4 * b + j];
    f32x4 acc = {0, 0, 0, 0};
#define STEP(B)                                                                                    \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                  \
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av[j], w[(4 * B + j) * 64 + l], acc, 4, B, 0);
    STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
    STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
    for (int r = 0; r < 4; ++r) out[r * 64 + l] = acc[r];
}

template <int NACC>
__global__ void __launch_bounds__(512) rate(int iters, float *out)
{
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = (f32x4){0, 0, 0, 0};
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16 / NACC; ++j)
#pragma unroll
            for (int n = 0; n < NACC; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, acc[n], 4, 5, 0);
    }
    float r = 0;
    for (int n = 0; n < NACC; ++n) r += acc[n][0];
    if (r == 12345.f) out[threadIdx.x] = r;
}

__global__ void __launch_bounds__(512) rate16(int iters, float *out)
{
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
        }
    }
    float r = a0[0] + a1[1];
    if (r == 12345.f) out[threadIdx.x] = r;
}

int main()
{
    float *a, *w, *o, ha[256], hw[4096], ho[256];
    hipMalloc(&a, 1024);
    hipMalloc(&w, 16384);
    hipMalloc(&o, 4096);
    srand(3);
    for (float &v : ha) v = (rand() % 2001 - 1000) * 1e-3f;
    for (float &v : hw) v = (rand() % 2001 - 1000) * 1e-3f;
    hipMemcpy(a, ha, 1024, hipMemcpyHostToDevice);
    hipMemcpy(w, hw, 16384, hipMemcpyHostToDevice);
    sem<<<1, 64>>>(a, w, o);
    hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 4; ++i)
        for (int l = 0; l < 64; ++l) {
            double s = 0;
            for (int k = 0; k < 64; ++k) s += (double)ha[i * 64 + k] * hw[k * 64 + l];
            worst = fmax(worst, fabs(s - ho[i * 64 + l]));
        }
    printf("semantics: max |err| = %.3g (%s)\n", worst, worst < 1e-4 ? "ok" : "WRONG");

    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 100000; // x 16 instructions (4x4x1) or 4 (16x16x4): 128 cycles per iteration
    auto run = [&](const char *name, auto launch) {
        launch(100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        launch(iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %.3f ms  -> %.1f cycles per 128-cycle block at 2.4 GHz\n", name, ms,
               ms * 1e-3 * 2.4e9 / iters);
    };
    run("4x4x1 1 acc, 1 wave/SIMD", [&](int n) { rate<1><<<256, 256>>>(n, o); });
    run("4x4x1 2 acc, 1 wave/SIMD", [&](int n) { rate<2><<<256, 256>>>(n, o); });
    run("4x4x1 4 acc, 1 wave/SIMD", [&](int n) { rate<4><<<256, 256>>>(n, o); });
    run("4x4x1 1 acc, 2 waves/SIMD (x2 work)", [&](int n) { rate<1><<<256, 512>>>(n, o); });
    run("4x4x1 2 acc, 2 waves/SIMD (x2 work)", [&](int n) { rate<2><<<256, 512>>>(n, o); });
    run("16x16x4 2 acc, 1 wave/SIMD", [&](int n) { rate16<<<256, 256>>>(n, o); });
    run("16x16x4 2 acc, 2 waves/SIMD (x2 work)", [&](int n) { rate16<<<256, 512>>>(n, o); });
    return 0;
}
