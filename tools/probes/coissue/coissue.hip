// Do the f64 / f32 16x16x4 MFMAs of gfx950 execute BESIDE vector instructions of another wave on the same SIMD, or do they
// share the vector pipe?  512-thread workgroups (two waves per SIMD), one per CU: waves 0-3 run an MFMA loop, waves 4-7 a VALU
// loop (f64 FMA / packed f32 FMA / scalar f32 FMA), each alone and both together.  co-issue: t(both) ~ max; shared: ~ sum.
// Build: hipcc --offload-arch=gfx950 -O3 coissue.hip -o coissue ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = __attribute__((ext_vector_type(4))) double;
using f4 = __attribute__((ext_vector_type(4))) float;
using f2 = __attribute__((ext_vector_type(2))) float;

template <int MF, int VA>   // MF: 0 none, 1 f64 mfma, 2 f32 mfma; VA: 0 none, 1 f64 fma, 2 pk f32 fma, 3 scalar f32 fma
__global__ void __launch_bounds__(512) k(double* out, int it_m, int it_v, double a0, double b0) {
    const int wave = threadIdx.x >> 6;
    double res = 0;
    if (wave < 4) {
        if constexpr (MF == 1) {
            d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            const double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
            for (int it = 0; it < it_m; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            }
            for (int i = 0; i < 4; ++i) res += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        } else if constexpr (MF == 2) {
            f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            const float a = (float)a0 + threadIdx.x, b = (float)b0 - threadIdx.x;
            for (int it = 0; it < it_m; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            }
            for (int i = 0; i < 4; ++i) res += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        }
    } else {
        if constexpr (VA == 1) {
            double x[8];
            for (int i = 0; i < 8; ++i) x[i] = a0 + i + threadIdx.x;
            const double m = 1.0 + b0 * 1e-9, c = b0 * 1e-12;
            for (int it = 0; it < it_v; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], m, c);
            }
            for (int i = 0; i < 8; ++i) res += x[i];
        } else if constexpr (VA == 2) {
            f2 x[8];
            for (int i = 0; i < 8; ++i) x[i] = (f2){(float)a0 + i, (float)b0 + threadIdx.x};
            const f2 m = {1.0f + (float)b0 * 1e-6f, 1.0f - (float)b0 * 1e-6f}, c = {(float)b0 * 1e-7f, (float)a0 * 1e-7f};
            for (int it = 0; it < it_v; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], m, c);
            }
            for (int i = 0; i < 8; ++i) res += x[i].x + x[i].y;
        } else if constexpr (VA == 3) {
            float x[8];
            for (int i = 0; i < 8; ++i) x[i] = (float)a0 + i + threadIdx.x;
            const float m = 1.0f + (float)b0 * 1e-6f, c = (float)b0 * 1e-7f;
            for (int it = 0; it < it_v; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(c));   // (plain C gets SLP-packed)
            }
            for (int i = 0; i < 8; ++i) res += x[i];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
}

template <int MF, int VA>
float run(double* out, int it_m, int it_v) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int pass = 0; pass < 3; ++pass) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MF, VA>), dim3(256), dim3(512), 0, 0, out, it_m, it_v, 1.0, 2.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (pass && ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best;
}

int main() {
    double* out;
    hipMalloc((void**)&out, 8 * 512 * 256);
    const int IM = 20000;        // 80 000 MFMAs per wave
    printf("| matrix loop (waves 0-3) | vector loop (waves 4-7) | matrix alone ms | vector alone ms | both ms | both / max | both / sum |\n|---|---|---|---|---|---|---|\n");
#define ROW(MF, VA, IV, mname, vname)                                                                                  \
    {                                                                                                                  \
        const float tm = run<MF, 0>(out, IM, 0), tv = run<0, VA>(out, 0, IV), tb = run<MF, VA>(out, IM, IV);           \
        printf("| %s | %s | %.3f | %.3f | %.3f | %.2f | %.2f |\n", mname, vname, tm, tv, tb, tb / (tm > tv ? tm : tv), tb / (tm + tv)); \
    }
    // vector iteration counts chosen so that each loop alone takes about as long as the f64 matrix loop
    ROW(1, 1, 160000, "v_mfma_f64_16x16x4_f64", "v_fma_f64 (8 chains)")
    ROW(1, 2, 160000, "v_mfma_f64_16x16x4_f64", "v_pk_fma_f32 (8 chains)")
    ROW(1, 3, 320000, "v_mfma_f64_16x16x4_f64", "v_fma_f32 (8 chains)")
    ROW(2, 1, 60000, "v_mfma_f32_16x16x4_f32", "v_fma_f64 (8 chains)")
    ROW(2, 2, 60000, "v_mfma_f32_16x16x4_f32", "v_pk_fma_f32 (8 chains)")
    ROW(2, 3, 120000, "v_mfma_f32_16x16x4_f32", "v_fma_f32 (8 chains)")
    return 0;
}
