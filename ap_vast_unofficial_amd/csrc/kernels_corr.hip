// K5': batched Hermitian rank-M correlation, stand-alone form (R goes to HBM).
//   R_B[k] = X_B[k]^H X_B[k], R_D[k] = X_D[k]^H X_D[k], r[k] = X_B[k]^H d[k]
// (complex twin of `R += Y @ Y.T`, `r += Y @ d`, reference Python/apvast.py:339-340, 347).
// The fused update kernels never write R to HBM; these entry points serve the apv_corr_dev / apv_gevd_vast_dev
// split of the C ABI, the stage-level parity tests, and the fp32-vs-bf16 accumulation study of BASELINE config 5.
//
//   corr_kernel<T>            any L <= 64, any M: LDS-staged VALU (f32 or f64 accumulation)
//   corr_mfma_f64_kernel      L in {16, 32, 48, 64}: v_mfma_f64_16x16x4_f64, float64 accumulation, one wave per 16x16 tile
//   corr_mfma_f32_kernel      L in {32, 64}: v_mfma_f32_32x32x2_f32, exact f32 products, one wave per (bin, matrix)
//   corr_mfma_bf16_kernel     L in {32, 64}: v_mfma_f32_32x32x16_bf16 on bf16 inputs, f32 accumulation
//
// In both MFMA kernels the slab is read ONCE from HBM straight into the operand registers: the two complex values a
// lane loads per row pair (columns l&31 and 32 + (l&31)) are the A (X^H) and the B (X) operands of all four 32x32
// tiles of R, so there is no LDS staging and the kernel is bound by HBM, not by the matrix cores.
#include "apv_internal.h"

namespace {

// XT: element type of the slabs (float2 = c64; double2 = c128, the float64 streaming front-end)
template <typename T, typename XT = float2>
__global__ void __launch_bounds__(256) corr_kernel(int M, int L, const XT* __restrict__ XB,
                                                   const XT* __restrict__ XD, const XT* __restrict__ d,
                                                   T* __restrict__ RB, T* __restrict__ RD, T* __restrict__ r) {
    constexpr int TPB = 256;
    constexpr int MT = 8;
    constexpr int NACC = (APV_MAX_N * APV_MAX_N) / TPB;     // 16
    __shared__ XT sX[MT * APV_MAX_N];
    __shared__ XT sd[MT];
    const int tid = threadIdx.x;
    const int k = blockIdx.x;
    for (int which = 0; which < 2; ++which) {
        const XT* X = (which ? XD : XB) + (size_t)k * M * L;
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
                        const XT xi = sX[m * L + i], xj = sX[m * L + j];
                        ax[a] += (T)xi.x * (T)xj.x + (T)xi.y * (T)xj.y;
                        ay[a] += (T)xi.x * (T)xj.y - (T)xi.y * (T)xj.x;
                    }
                }
            }
            if (which == 0 && tid < L) {
                for (int m = 0; m < rows; ++m) {
                    const XT xi = sX[m * L + tid], dm = sd[m];
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

using d4c = __attribute__((ext_vector_type(4))) double;

// float64 accumulation on v_mfma_f64_16x16x4_f64: one workgroup per (bin, matrix), one wave per 16 x 16 tile of R
// (NT = L / 16 tiles per side).  Lane (c = lane & 15, h = lane >> 4) supplies A[i = c][k = h] = conj(X[m0+h][16 ta + c])
// and B[k = h][j = c] = X[m0+h][16 tb + c]; the products of float32 inputs are exact in double.  Rows are taken eight
// k-steps (32 control points) at a time with every load of the chunk in flight before its first MFMA.
template <int NT>
__global__ void __launch_bounds__(64 * NT * NT) corr_mfma_f64_kernel(int M, const float2* __restrict__ XB,
                                                                     const float2* __restrict__ XD,
                                                                     const float2* __restrict__ d, double2* __restrict__ RB,
                                                                     double2* __restrict__ RD, double2* __restrict__ r) {
    constexpr int L = 16 * NT;
    const int k = blockIdx.x, which = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ta = wave / NT, tb = wave % NT;
    const float2* X = (which ? XD : XB) + (size_t)k * M * L;
    const int c = lane & 15, h = lane >> 4;
    const bool want_r = (which == 0) && (tb == 0);          // the waves of tile column 0 cover every loudspeaker once
    d4c re = {0, 0, 0, 0}, im = {0, 0, 0, 0};
    double rx = 0.0, ry = 0.0;
    for (int mc = 0; mc < M; mc += 32) {
        float2 xa[8], xb[8], dv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int m = mc + 4 * q + h;
            const bool ok = m < M;
            xa[q] = ok ? X[(size_t)m * L + 16 * ta + c] : make_float2(0.f, 0.f);
            xb[q] = ok ? X[(size_t)m * L + 16 * tb + c] : make_float2(0.f, 0.f);
            dv[q] = (ok && want_r) ? d[(size_t)k * M + m] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double ar = xa[q].x, ai = xa[q].y, br = xb[q].x, bi = xb[q].y;
            re = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, re, 0, 0, 0);
            re = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, bi, re, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, im, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, br, im, 0, 0, 0);
            rx += ar * (double)dv[q].x + ai * (double)dv[q].y;
            ry += ar * (double)dv[q].y - ai * (double)dv[q].x;
        }
    }
    // f64 16x16x4 accumulator: row = (lane >> 4) + 4 t, col = lane & 15
    double2* R = (which ? RD : RB) + (size_t)k * L * L;
#pragma unroll
    for (int t = 0; t < 4; ++t) R[(size_t)(16 * ta + h + 4 * t) * L + 16 * tb + c] = make_double2(re[t], im[t]);
    if (want_r) {
        rx += __shfl_xor(rx, 16, 64); ry += __shfl_xor(ry, 16, 64);
        rx += __shfl_xor(rx, 32, 64); ry += __shfl_xor(ry, 32, 64);
        if (h == 0) r[(size_t)k * L + 16 * ta + c] = make_double2(rx, ry);
    }
}

using f16v = __attribute__((ext_vector_type(16))) float;
using bf8v = __attribute__((ext_vector_type(8))) short;

// accumulator element r of a 32x32 tile: row = (r&3) + 8 (r>>2) + 4 (lane>>5), col = lane & 31
__device__ __forceinline__ void store_tile(float2* __restrict__ R, int L, int ti, int tj, const f16v& re, const f16v& im,
                                           int lane) {
    const int col = tj * 32 + (lane & 31), h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        R[(size_t)row * L + col] = make_float2(re[r], im[r]);
    }
}

// One workgroup per (bin, matrix), one wave per 32x32 tile of R (NT = L / 32 tiles per side): wave w owns tile
// (a, b) = (w / NT, w % NT) and needs only the column halves a and b of every row.
template <int NT>
__global__ void __launch_bounds__(64 * NT * NT) corr_mfma_f32_kernel(int M, const float2* __restrict__ XB,
                                                                     const float2* __restrict__ XD,
                                                                     const float2* __restrict__ d,
                                                                     float2* __restrict__ RB, float2* __restrict__ RD,
                                                                     float2* __restrict__ r) {
    constexpr int L = 32 * NT;
    constexpr int G = 4;                                   // k-steps in flight
    const int k = blockIdx.x, which = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ta = wave / NT, tb = wave % NT;
    const float2* X = (which ? XD : XB) + (size_t)k * M * L;
    const int c = lane & 31, h = lane >> 5;
    f16v re, im;
#pragma unroll
    for (int q = 0; q < 16; ++q) { re[q] = 0.f; im[q] = 0.f; }
    float rx = 0.f, ry = 0.f;
    const bool want_r = (which == 0) && (tb == 0);          // the waves of tile column 0 cover every loudspeaker once
    // lane (c, h): A[i = c][k = h] = conj(X[m0 + h][32 ta + c]), B[k = h][j = c] = X[m0 + h][32 tb + c]
    const float2* pa = X + (size_t)h * L + ta * 32 + c;
    const float2* pb = X + (size_t)h * L + tb * 32 + c;
    const float2* pd = d + (size_t)k * M + h;
    const int steps = M / 2, groups = steps / G;             // M even is required (checked by the launcher)
    float2 na[G], nb[G], nd[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
        na[u] = pa[(size_t)(2 * u) * L];                     // M >= 2 G (launcher)
        nb[u] = pb[(size_t)(2 * u) * L];
        nd[u] = pd[2 * u];
    }
    for (int g = 0; g < groups; ++g) {
        float2 xa[G], xb[G], dv[G];
#pragma unroll
        for (int u = 0; u < G; ++u) { xa[u] = na[u]; xb[u] = nb[u]; dv[u] = nd[u]; }
        if (g + 1 < groups) {
            const size_t m0 = (size_t)(g + 1) * 2 * G;
#pragma unroll
            for (int u = 0; u < G; ++u) {
                na[u] = pa[(m0 + 2 * u) * L];
                nb[u] = pb[(m0 + 2 * u) * L];
                if (want_r) nd[u] = pd[m0 + 2 * u];
            }
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if (want_r) {
                rx += xa[u].x * dv[u].x + xa[u].y * dv[u].y;
                ry += xa[u].x * dv[u].y - xa[u].y * dv[u].x;
            }
            re = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[u].x, xb[u].x, re, 0, 0, 0);
            re = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[u].y, xb[u].y, re, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[u].x, xb[u].y, im, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f32_32x32x2f32(-xa[u].y, xb[u].x, im, 0, 0, 0);
        }
    }
    for (int st = groups * G; st < steps; ++st) {
        const float2 xa = pa[(size_t)(2 * st) * L], xb = pb[(size_t)(2 * st) * L];
        if (want_r) {
            const float2 dv = pd[2 * st];
            rx += xa.x * dv.x + xa.y * dv.y;
            ry += xa.x * dv.y - xa.y * dv.x;
        }
        re = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.x, xb.x, re, 0, 0, 0);
        re = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.y, xb.y, re, 0, 0, 0);
        im = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.x, xb.y, im, 0, 0, 0);
        im = __builtin_amdgcn_mfma_f32_32x32x2f32(-xa.y, xb.x, im, 0, 0, 0);
    }
    store_tile((which ? RD : RB) + (size_t)k * L * L, L, ta, tb, re, im, lane);
    if (want_r) {
        const float sx = rx + __shfl_xor(rx, 32, 64), sy = ry + __shfl_xor(ry, 32, 64);
        if (h == 0) r[(size_t)k * L + ta * 32 + c] = make_float2(sx, sy);
    }
}

// bf16 inputs: X as (re, im) bf16 pairs, 4 bytes per complex element, same [K][M][L] layout.
// v_mfma_f32_32x32x16_bf16: lane (c = l&31, h = l>>5) holds A[i = c][k = 8h + j], B[k = 8h + j][j' = c], j = 0..7.
template <int NT>
__global__ void __launch_bounds__(64 * NT * NT) corr_mfma_bf16_kernel(int M, const uint32_t* __restrict__ XB,
                                                                      const uint32_t* __restrict__ XD,
                                                                      const uint32_t* __restrict__ d,
                                                                      float2* __restrict__ RB, float2* __restrict__ RD,
                                                                      float2* __restrict__ r) {
    constexpr int L = 32 * NT;
    const int k = blockIdx.x, which = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ta = wave / NT, tb = wave % NT;
    const uint32_t* X = (which ? XD : XB) + (size_t)k * M * L;
    const int c = lane & 31, h = lane >> 5;
    f16v re, im;
#pragma unroll
    for (int q = 0; q < 16; ++q) { re[q] = 0.f; im[q] = 0.f; }
    float rx = 0.f, ry = 0.f;
    const bool want_r = (which == 0) && (tb == 0);
    const uint32_t* pa = X + (size_t)(8 * h) * L + ta * 32 + c;
    const uint32_t* pb = X + (size_t)(8 * h) * L + tb * 32 + c;
    const int steps = M / 16;                                // M % 16 == 0 is required (checked by the launcher)
    uint32_t na[8], nb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        na[j] = pa[(size_t)j * L];                           // M >= 16 (launcher)
        nb[j] = pb[(size_t)j * L];
    }
    for (int st = 0; st < steps; ++st) {
        uint32_t va[8], vb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { va[j] = na[j]; vb[j] = nb[j]; }
        if (st + 1 < steps) {
            const size_t m0 = (size_t)(st + 1) * 16;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                na[j] = pa[(m0 + j) * L];
                nb[j] = pb[(m0 + j) * L];
            }
        }
        bf8v ar, ai, nai, br, bi;
#pragma unroll
        for (int j = 0; j < 8; ++j) {                        // lo half = re, hi half = im
            ar[j] = (short)(va[j] & 0xffffu);
            ai[j] = (short)(va[j] >> 16);
            nai[j] = (short)((va[j] >> 16) ^ 0x8000u);
            br[j] = (short)(vb[j] & 0xffffu);
            bi[j] = (short)(vb[j] >> 16);
        }
        {   // r = X_B^H d: every wave accumulates (no divergent region in the loop), only tile column 0 stores
            const uint32_t* dp = d + (size_t)k * M + st * 16 + 8 * h;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t dv = dp[j];
                const float dr = __uint_as_float(dv << 16), di = __uint_as_float(dv & 0xffff0000u);
                const float fr = __uint_as_float(va[j] << 16), fi = __uint_as_float(va[j] & 0xffff0000u);
                rx += fr * dr + fi * di;
                ry += fr * di - fi * dr;
            }
        }
        re = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar, br, re, 0, 0, 0);
        re = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ai, bi, re, 0, 0, 0);
        im = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar, bi, im, 0, 0, 0);
        im = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nai, br, im, 0, 0, 0);
    }
    store_tile((which ? RD : RB) + (size_t)k * L * L, L, ta, tb, re, im, lane);
    const float sx = rx + __shfl_xor(rx, 32, 64), sy = ry + __shfl_xor(ry, 32, 64);
    if (want_r && h == 0) r[(size_t)k * L + ta * 32 + c] = make_float2(sx, sy);
}

// c64 -> (bf16, bf16), round to nearest even (plain cast: v_cvt_pk_bf16_f32 keeps NaNs NaN)
__global__ void __launch_bounds__(256) to_bf16_kernel(size_t count, const float2* __restrict__ in, uint32_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float2 v = in[i];
    const __bf16 a = (__bf16)v.x, b = (__bf16)v.y;
    const uint16_t ua = *reinterpret_cast<const uint16_t*>(&a), ub = *reinterpret_cast<const uint16_t*>(&b);
    out[i] = (uint32_t)ua | ((uint32_t)ub << 16);
}

}  // namespace

hipError_t apv_launch_corr(int compute_dtype, int K, int M, int L, const float2* XB, const float2* XD,
                           const float2* d, void* RB, void* RD, void* r, hipStream_t s) {
    if (K <= 0) return hipSuccess;
    if (L < 1 || L > APV_MAX_N || M < 1) return hipErrorInvalidValue;
    if (compute_dtype == APV_F64 && (L == 16 || L == 32 || L == 48 || L == 64)) {
        double2 *rb = (double2*)RB, *rd = (double2*)RD, *rr = (double2*)r;
        if (L == 16) hipLaunchKernelGGL(corr_mfma_f64_kernel<1>, dim3(K, 2), dim3(64), 0, s, M, XB, XD, d, rb, rd, rr);
        else if (L == 32) hipLaunchKernelGGL(corr_mfma_f64_kernel<2>, dim3(K, 2), dim3(256), 0, s, M, XB, XD, d, rb, rd, rr);
        else if (L == 48) hipLaunchKernelGGL(corr_mfma_f64_kernel<3>, dim3(K, 2), dim3(576), 0, s, M, XB, XD, d, rb, rd, rr);
        else hipLaunchKernelGGL(corr_mfma_f64_kernel<4>, dim3(K, 2), dim3(1024), 0, s, M, XB, XD, d, rb, rd, rr);
    } else if (compute_dtype == APV_F64) {
        hipLaunchKernelGGL(corr_kernel<double>, dim3(K), dim3(256), 0, s, M, L, XB, XD, d, (double*)RB,
                           (double*)RD, (double*)r);
    } else if (L == 64 && (M % 2) == 0 && M >= 8) {
        hipLaunchKernelGGL(corr_mfma_f32_kernel<2>, dim3(K, 2), dim3(256), 0, s, M, XB, XD, d, (float2*)RB, (float2*)RD, (float2*)r);
    } else if (L == 32 && (M % 2) == 0 && M >= 8) {
        hipLaunchKernelGGL(corr_mfma_f32_kernel<1>, dim3(K, 2), dim3(64), 0, s, M, XB, XD, d, (float2*)RB, (float2*)RD, (float2*)r);
    } else {
        hipLaunchKernelGGL(corr_kernel<float>, dim3(K), dim3(256), 0, s, M, L, XB, XD, d, (float*)RB,
                           (float*)RD, (float*)r);
    }
    return hipGetLastError();
}

// float64 statistics from c128 slabs (on-demand attributes of the float64 streaming front-end)
hipError_t apv_launch_corr_c128(int K, int M, int L, const double2* XB, const double2* XD, const double2* d, double2* RB,
                                double2* RD, double2* r, hipStream_t s) {
    if (K <= 0) return hipSuccess;
    if (L < 1 || L > APV_MAX_N || M < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL((corr_kernel<double, double2>), dim3(K), dim3(256), 0, s, M, L, XB, XD, d, (double*)RB, (double*)RD, (double*)r);
    return hipGetLastError();
}

hipError_t apv_launch_corr_bf16(int K, int M, int L, const uint32_t* XB, const uint32_t* XD, const uint32_t* d,
                                float2* RB, float2* RD, float2* r, hipStream_t s) {
    if (K <= 0) return hipSuccess;
    if ((M % 16) != 0 || M < 16) return hipErrorInvalidValue;
    if (L == 64) hipLaunchKernelGGL(corr_mfma_bf16_kernel<2>, dim3(K, 2), dim3(256), 0, s, M, XB, XD, d, RB, RD, r);
    else if (L == 32) hipLaunchKernelGGL(corr_mfma_bf16_kernel<1>, dim3(K, 2), dim3(64), 0, s, M, XB, XD, d, RB, RD, r);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t apv_launch_to_bf16(size_t count, const float2* in, uint32_t* out, hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, count, in, out);
    return hipGetLastError();
}
