// Issue cost of the instruction classes the order-16 kernel is made of, per SIMD, at 1 / 2 / 4 waves per SIMD, in CYCLES of the
// clock the chip actually holds in that loop (s_memtime / s_memrealtime stamps around the loop of one wave).
// Build: hipcc --offload-arch=gfx950 -O3 issue_cost.hip -o issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = __attribute__((ext_vector_type(4))) double;
using f4 = __attribute__((ext_vector_type(4))) float;
using f2 = __attribute__((ext_vector_type(2))) float;

enum { FMA64 = 0, PKFMA32, FMA32, MFMA64, MFMA32, DPPADD32, RSQ32, MOVDPP, NOPS };
static const char* kNames[] = {"v_fma_f64", "v_pk_fma_f32", "v_fma_f32", "v_mfma_f64_16x16x4_f64", "v_mfma_f32_16x16x4_f32",
                               "v_add_f32 dpp quad_perm", "v_rsq_f32", "v_mov_b32 dpp row_ror:8"};

template <int OP>
__global__ void __launch_bounds__(1024) k(double* out, unsigned long long* stamps, int iters, double a0, double b0) {
    double res = 0;
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    if constexpr (OP == FMA64) {
        double x[8];
        for (int i = 0; i < 8; ++i) x[i] = a0 + i + threadIdx.x;
        const double m = 1.0 + b0 * 1e-9, c = b0 * 1e-12;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], m, c);
        }
        for (int i = 0; i < 8; ++i) res += x[i];
    } else if constexpr (OP == PKFMA32) {
        f2 x[8];
        for (int i = 0; i < 8; ++i) x[i] = (f2){(float)a0 + i, (float)b0 + threadIdx.x};
        const f2 m = {1.0f + (float)b0 * 1e-6f, 1.0f - (float)b0 * 1e-6f}, c = {(float)b0 * 1e-7f, (float)a0 * 1e-7f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], m, c);
        }
        for (int i = 0; i < 8; ++i) res += x[i].x + x[i].y;
    } else if constexpr (OP == FMA32 || OP == DPPADD32 || OP == RSQ32 || OP == MOVDPP) {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = (float)a0 + i + threadIdx.x;
        const float m = 1.0f + (float)b0 * 1e-6f, c = (float)b0 * 1e-7f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(c));
                else if constexpr (OP == DPPADD32) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
                else if constexpr (OP == RSQ32) asm volatile("v_rsq_f32 %0, %0" : "+v"(x[i]));
                else asm volatile("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
            }
        }
        for (int i = 0; i < 8; ++i) res += x[i];
    } else if constexpr (OP == MFMA64) {
        d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        const double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 3], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) res += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if constexpr (OP == MFMA32) {
        f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        const float a = (float)a0 + threadIdx.x, b = (float)b0 - threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 3], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) res += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        // the stamps go to a buffer of their own (never into an output the kernel's results depend on)
        asm volatile("s_waitcnt lgkmcnt(0)");
        stamps[0] = __builtin_amdgcn_s_memtime() - t0;
        stamps[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = res;
}

template <int OP>
void run(double* out, unsigned long long* stamps, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 4}) {                    // waves per SIMD: 256 / 512 / 1024-thread workgroups, one per CU
        float best = 1e30f;
        unsigned long long st[2] = {0, 0};
        // 40 launches back to back: the clock the loop settles at, not the boost clock of a cold start
        for (int pass = 0; pass < 40; ++pass) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256 * wps), 0, 0, out, stamps, iters, 1.0, 2.0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (pass >= 30 && ms < best) { best = ms; hipMemcpy(st, stamps, 16, hipMemcpyDeviceToHost); }
        }
        const double n_per_simd = (double)iters * 8 * wps;
        const double ghz = st[1] ? (double)st[0] / ((double)st[1] * 10.0) : 0.0;     // s_memrealtime ticks at 100 MHz
        printf("| %s | %d | %.3f | %.2f | %.2f | %.1f |\n", kNames[OP], wps, best, best * 1e6 / n_per_simd, ghz, best * 1e6 / n_per_simd * ghz);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    double* out;
    unsigned long long* stamps;
    hipMalloc((void**)&out, 8 * 1024 * 256);
    hipMalloc((void**)&stamps, 64);
    printf("| instruction | waves / SIMD | ms | ns per instruction per SIMD | in-kernel clock GHz | cycles per instruction per SIMD |\n|---|---|---|---|---|---|\n");
    run<FMA64>(out, stamps, 20000);
    run<PKFMA32>(out, stamps, 20000);
    run<FMA32>(out, stamps, 20000);
    run<DPPADD32>(out, stamps, 20000);
    run<MOVDPP>(out, stamps, 20000);
    run<RSQ32>(out, stamps, 20000);
    run<MFMA64>(out, stamps, 4000);
    run<MFMA32>(out, stamps, 4000);
    return 0;
}
