// Issue rate of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 on gfx950: independent accumulators, operands in registers,
// no memory traffic.  Build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = __attribute__((ext_vector_type(4))) double;
using f4 = __attribute__((ext_vector_type(4))) float;

template <int NACC>
__global__ void __launch_bounds__(256) k_f64(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    const double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k_f32(float* out, int iters, float a0, float b0) {
    f4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
    const float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    void* out;
    hipMalloc(&out, 8 * 256 * 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int wgs_per_cu : {1, 2, 4, 8}) {
        const int grid = 256 * wgs_per_cu;                 // 4 waves per workgroup: 1 or 2 waves per SIMD
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_f64<4>, dim3(grid), dim3(256), 0, 0, (double*)out, iters, 1.0, 2.0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double mf = (double)grid * 4 * iters * 4;      // MFMAs
            if (pass) printf("f64 16x16x4, %d wave(s)/SIMD: %.3f ms, %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n", wgs_per_cu, ms,
                             ms * 1e6 / (mf / 1024), mf * 2048 / ms / 1e9);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_f32<4>, dim3(grid), dim3(256), 0, 0, (float*)out, iters, 1.0f, 2.0f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            if (pass) printf("f32 16x16x4, %d wave(s)/SIMD: %.3f ms, %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n", wgs_per_cu, ms,
                             ms * 1e6 / (mf / 1024), mf * 2048 / ms / 1e9);
        }
    }
    return 0;
}
