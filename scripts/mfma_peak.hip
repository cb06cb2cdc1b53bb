// Measures the achievable v_mfma_f64_16x16x4_f64 and v_mfma_f32_32x32x2_f32 rates on this GPU
// (back-to-back issue, independent accumulators, W waves per SIMD).  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void f64_loop(double* out, int iters) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3 + 1.0, b = blockIdx.x * 1e-3 + 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void f32_loop(float* out, int iters) {
    v16f acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    float a = threadIdx.x * 1e-3f + 1.0f, b = blockIdx.x * 1e-3f + 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* o; hipMalloc(&o, 1 << 26);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = 256 * 4, threads = 256 * wps / 1;  // 4 WG per CU x (wps) ... total waves/CU = 16*wps
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(f64_loop<4>, dim3(256 * wps), dim3(256), 0, 0, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fl = 2.0 * 16 * 16 * 4 * 4.0 * iters * (256.0 * wps * 4);
            if (rep) printf("f64 16x16x4: %d wave/SIMD: %.3f ms  %.1f TFLOP/s\n", wps, ms, fl / ms / 1e9);
        }
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(f32_loop<4>, dim3(256 * wps), dim3(256), 0, 0, (float*)o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fl = 2.0 * 32 * 32 * 2 * 4.0 * iters * (256.0 * wps * 4);
            if (rep) printf("f32 32x32x2: %d wave/SIMD: %.3f ms  %.1f TFLOP/s\n", wps, ms, fl / ms / 1e9);
        }
    }
    return 0;
}
