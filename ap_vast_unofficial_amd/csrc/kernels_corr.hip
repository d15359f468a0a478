// K5': batched Hermitian rank-M correlation, stand-alone form.
//   R_B[k] = X_B[k]^H X_B[k], R_D[k] = X_D[k]^H X_D[k], r[k] = X_B[k]^H d[k]
// (complex twin of `R += Y @ Y.T`, `r += Y @ d`, reference Python/apvast.py:339-340, 347).
// The fused update kernels never write R to HBM; this entry point exists for the
// apv_corr_dev / apv_gevd_vast_dev split of the C ABI and for stage-level parity tests.
#include "apv_internal.h"

namespace {

template <typename T>
__global__ void __launch_bounds__(256) corr_kernel(int M, int L, const float2* __restrict__ XB,
                                                   const float2* __restrict__ XD, const float2* __restrict__ d,
                                                   T* __restrict__ RB, T* __restrict__ RD, T* __restrict__ r) {
    constexpr int TPB = 256;
    constexpr int MT = 8;
    constexpr int NACC = (APV_MAX_N * APV_MAX_N) / TPB;     // 16
    __shared__ float2 sX[MT * APV_MAX_N];
    __shared__ float2 sd[MT];
    const int tid = threadIdx.x;
    const int k = blockIdx.x;
    for (int which = 0; which < 2; ++which) {
        const float2* X = (which ? XD : XB) + (size_t)k * M * L;
        T ax[NACC], ay[NACC];
#pragma unroll
        for (int a = 0; a < NACC; ++a) ax[a] = ay[a] = 0;
        T rx = 0, ry = 0;
        for (int m0 = 0; m0 < M; m0 += MT) {
            const int rows = (M - m0) < MT ? (M - m0) : MT;
            for (int idx = tid; idx < rows * L; idx += TPB) sX[idx] = X[(size_t)m0 * L + idx];
            if (which == 0 && tid < rows) sd[tid] = d[(size_t)k * M + m0 + tid];
            __syncthreads();
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                const int idx = tid + a * TPB;
                if (idx < L * L) {
                    const int i = idx / L, j = idx - i * L;
                    for (int m = 0; m < rows; ++m) {
                        const float2 xi = sX[m * L + i], xj = sX[m * L + j];
                        ax[a] += (T)xi.x * (T)xj.x + (T)xi.y * (T)xj.y;
                        ay[a] += (T)xi.x * (T)xj.y - (T)xi.y * (T)xj.x;
                    }
                }
            }
            if (which == 0 && tid < L) {
                for (int m = 0; m < rows; ++m) {
                    const float2 xi = sX[m * L + tid], dm = sd[m];
                    rx += (T)xi.x * (T)dm.x + (T)xi.y * (T)dm.y;
                    ry += (T)xi.x * (T)dm.y - (T)xi.y * (T)dm.x;
                }
            }
            __syncthreads();
        }
        T* R = (which ? RD : RB) + (size_t)k * L * L * 2;
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            const int idx = tid + a * TPB;
            if (idx < L * L) {
                R[2 * idx] = ax[a];
                R[2 * idx + 1] = ay[a];
            }
        }
        if (which == 0 && tid < L) {
            r[((size_t)k * L + tid) * 2] = rx;
            r[((size_t)k * L + tid) * 2 + 1] = ry;
        }
    }
}

}  // namespace

hipError_t apv_launch_corr(int compute_dtype, int K, int M, int L, const float2* XB, const float2* XD,
                           const float2* d, void* RB, void* RD, void* r, hipStream_t s) {
    if (K <= 0) return hipSuccess;
    if (L < 1 || L > APV_MAX_N || M < 1) return hipErrorInvalidValue;
    if (compute_dtype == APV_F64)
        hipLaunchKernelGGL(corr_kernel<double>, dim3(K), dim3(256), 0, s, M, L, XB, XD, d, (double*)RB,
                           (double*)RD, (double*)r);
    else
        hipLaunchKernelGGL(corr_kernel<float>, dim3(K), dim3(256), 0, s, M, L, XB, XD, d, (float*)RB,
                           (float*)RD, (float*)r);
    return hipGetLastError();
}
