// Joint diagonalisation of REAL symmetric pairs of broadband order (n = filter_length x loudspeakers: 256 at
// cfg1, 800 with the parameters of make_python_test.m), float64, matrices resident in HBM/L2.
//
//   jdiag(A, B)   reference Python/apvast.py:20-36, called at apvast.py:380, 382 with n = J L
//
//   1  B + reg I = L L^T               left-looking Cholesky in 32-wide column panels, one launch per panel; the
//                                      32 x 32 diagonal block is eliminated on [D | I] in LDS, which also yields
//                                      its inverse                                            apvast.py:22-27
//      W = L^-1                        forward substitution on 32 x 32 tiles, one launch
//   2  C = W A W^T                     two tiled GEMMs                                         apvast.py:28-29
//   3  C = Q diag(lam) Q^T             block cyclic Jacobi: the order is cut into 16-wide blocks, paired round-robin;
//                                      one launch per block round.  A workgroup owns one 32 x 32 tile (P, Q) of the
//                                      pair grid, re-derives the rotations of BOTH diagonal tiles it depends on with
//                                      an in-LDS Jacobi sweep (redundant across the row/column of workgroups, but it
//                                      removes every dependency inside the launch), and applies V_P^T . V_Q.
//                                      X = W^T Q is accumulated in place by the same workgroups.    apvast.py:30-35
//   4  rank of every eigenvalue (descending), columns of X gathered in that order               apvast.py:31-35
//   5  (optional) w_v = sum_{i<v} (x_i^T r)/(lam_i + mu) x_i for v = 1..V                       apvast.py:406-414
//
// The sequential depth is what bounds a single n = 256 problem, not flops or bytes: ~8 sweeps x (n/16 - 1)
// launches, each dominated by 16 (31 in the first round of a sweep) LDS-synchronised inner rounds.
#include "apv_internal.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

namespace {

constexpr int TPB = 256;

constexpr int BT = 32;   // tile edge: two 16-wide Jacobi blocks, one Cholesky panel
constexpr int BH = 16;
constexpr int LS = 33;   // LDS row stride in doubles (bank-conflict padding)

// 32 x 32 x 32 product from LDS tiles, 2 x 2 outputs per thread: o = {(ty,tx), (ty,tx+16), (ty+16,tx), (ty+16,tx+16)}
// of sum_k A(t,k) B(k,u) with A(t,k) = TA ? A[k][t] : A[t][k], B(k,u) = TB ? B[u][k] : B[k][u]
using d4 = __attribute__((ext_vector_type(4))) double;

// The same product on v_mfma_f64_16x16x4_f64: four waves, wave w owns the 16 x 16 tile (w >> 1, w & 1).
// acc[t] is element (row0 + (lane >> 4) + 4 t, col0 + (lane & 15)).
template <bool TA, bool TB = false>
__device__ __forceinline__ d4 mm32_mfma(const double* A, const double* B, int w, int lane) {
    const int row0 = (w >> 1) * 16, col0 = (w & 1) * 16, il = lane & 15, kq = lane >> 4;
    d4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int k0 = 0; k0 < BT; k0 += 4) {
        const double av = TA ? A[(k0 + kq) * LS + row0 + il] : A[(row0 + il) * LS + k0 + kq];
        const double bv = TB ? B[(col0 + il) * LS + k0 + kq] : B[(k0 + kq) * LS + col0 + il];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
    return acc;
}

// 1/sqrt(x) and 1/x to full double precision without the division sequence: v_rsq_f64 / v_rcp_f64 (5e-8 on gfx950) and Newton steps
__device__ __forceinline__ double pn_rsq(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
}
__device__ __forceinline__ double pn_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = y * __builtin_fma(-x, y, 2.0);
    return y * __builtin_fma(-x, y, 2.0);
}

// ONE wave: D (order 16, symmetric, in LDS with row stride LS) -> W = L^-1 with D = L L^T, written to Wout (lower triangle, zeros
// above).  [D | I] is eliminated with the rows in registers: lane = row i + 16 x column group cq holds D[i][4 cq ..] and the right
// half's [i][4 cq ..]; a step hands the pivot COLUMN of the left half (its pivot row, by symmetry of what is left) and the pivot row of
// the right half round through 32 doubles of LDS -- no barrier: the LDS executes a wave's accesses in order (the scheme of the
// order-64 kernel's factorisation, kernels_gevd64.hip stage 1).  Returns true when a pivot is not positive and finite.
__device__ __forceinline__ bool wave_inv_chol16(const double* D, double* Wout, double* buf, int lane) {
    const int i = lane & 15, cq = lane >> 4;
    double b[4], w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        b[u] = D[i * LS + 4 * cq + u];
        w[u] = (4 * cq + u == i) ? 1.0 : 0.0;
    }
    double dsc = 1.0;
    bool bad = false;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int qc = q >> 2, qu = q & 3;
        if (cq == qc) buf[i] = b[qu];
        if (i == q) {
#pragma unroll
            for (int u = 0; u < 4; ++u) buf[16 + 4 * cq + u] = w[u];
        }
        const double dq = buf[q];
        bad = bad || !(dq > 0.0) || !(dq < 1e300);
        const double inv = pn_rcp(dq);
        if (i == q) dsc = pn_rsq(dq);
        if (i > q) {
            const double m = buf[i] * inv;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                b[u] = __builtin_fma(-m, buf[4 * cq + u], b[u]);
                w[u] = __builtin_fma(-m, buf[16 + 4 * cq + u], w[u]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) Wout[i * LS + 4 * cq + u] = (4 * cq + u <= i) ? w[u] * dsc : 0.0;
    return bad;
}

// one wave, 16 x 16 x 16 on v_mfma_f64_16x16x4_f64 with operands fetched by fa(i, k), fb(k, j): acc[t] = element (kq + 4 t, il)
template <typename FA, typename FB>
__device__ __forceinline__ d4 wave_mm16(FA fa, FB fb, int lane) {
    const int il = lane & 15, kq = lane >> 4;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa(il, k0 + kq), fb(k0 + kq, il), acc, 0, 0, 0);
    return acc;
}

// ---- step 1 ---------------------------------------------------------------------------------------
// Left-looking Cholesky, column panel k (32 wide); workgroup x handles the row tile I = k + x.  Every workgroup
// forms the updated diagonal block D = B[K,K] - sum_J L[K,J] L[K,J]^T itself and eliminates [D | I] in LDS (unscaled
// columns of L_D on the left, rows of L_D^-1 up to 1/sqrt(d) on the right), then writes L[I,K] = T L_D^-T.
// The strictly lower tiles of B are replaced by L; LiBuf[k] receives L_D^-1 (L_D itself is not kept).
// diagnostics (APV_LARGE_DEBUG=2): s_memtime of thread 0 of the LAST row tile's workgroup of matrix 0 at the phase boundaries, [panel][4]
__device__ unsigned long long g_panel_stamps[64 * 4];
__device__ int g_panel_stamps_on;

template <bool WAVE_ELIM>
__global__ void __launch_bounds__(256) chol_panel_kernel(int ld, int k, int nbk, double* __restrict__ B,
                                                         double* __restrict__ LiBuf, int* __restrict__ flag,
                                                         size_t mat_stride) {
    __shared__ double La[BT * LS], Lb[BT * LS], Dm[BT * LS], Wp[BT * LS], rs[BT];
    __shared__ int sflag;
    const int z = blockIdx.z;
    // workgroup 0 of this very launch may raise the flag: read it once per workgroup, so that all waves take the same way
    if (threadIdx.x == 0) sflag = flag[z];
    __syncthreads();
    if (sflag) return;
    B += z * mat_stride;
    LiBuf += (size_t)z * nbk * BT * BT;
    const int I = k + blockIdx.x;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const bool stamping = g_panel_stamps_on && z == 0 && I == nbk - 1 && tid == 0 && k < 64;
    auto stamp = [&](int i) { if (stamping) g_panel_stamps[4 * k + i] = __builtin_amdgcn_s_memtime(); };
    stamp(0);
    // the products of the left-looking update on the f64 matrix cores, one 16 x 16 quadrant per wave (round 3: they ran on the
    // vector ALU with four LDS reads per four multiply-adds, 3 us per product; the late panels of n = 800 took 100-200 us)
    const int wq = tid >> 6, lq = tid & 63;
    d4 accT = {0.0, 0.0, 0.0, 0.0}, accD = {0.0, 0.0, 0.0, 0.0};
    // the tiles of term J + 1 are fetched into registers while the matrix cores work on term J (round 4: one term was a load, a
    // barrier, two products and a barrier, ~0.8 us, 300 of them at n = 800)
    double ra[BT * BT / 256], rb[BT * BT / 256];
    auto gload = [&](int J) {
#pragma unroll
        for (int e4 = 0; e4 < BT * BT / 256; ++e4) {
            const int e = tid + 256 * e4;
            const int t = e >> 5, u = e & 31;
            ra[e4] = B[(size_t)(I * BT + t) * ld + J * BT + u];
            rb[e4] = B[(size_t)(k * BT + t) * ld + J * BT + u];
        }
    };
    if (k > 0) gload(0);
    for (int J = 0; J < k; ++J) {
#pragma unroll
        for (int e4 = 0; e4 < BT * BT / 256; ++e4) {
            const int e = tid + 256 * e4;
            const int t = e >> 5, u = e & 31;
            La[t * LS + u] = ra[e4];
            Lb[t * LS + u] = rb[e4];
        }
        __syncthreads();
        if (J + 1 < k) gload(J + 1);
        const d4 oD = mm32_mfma<false, true>(Lb, Lb, wq, lq);
        for (int i = 0; i < 4; ++i) accD[i] -= oD[i];
        if (I != k) {
            const d4 oT = mm32_mfma<false, true>(La, Lb, wq, lq);
            for (int i = 0; i < 4; ++i) accT[i] -= oT[i];
        }
        __syncthreads();
    }
    // (Measured with the phase stamps below, APV_LARGE_DEBUG=2: a term of this loop costs ~2 050 cycles = 0.85 us, the diagonal block
    // 9.6 us, the last product and the stores 0.45 us.  Three other forms of the loop were tried at the end of round 4 -- two and
    // four terms per stage with the next stage's tiles in registers, the two accumulator chains of a wave interleaved and sharing
    // their B operand, a term's 24 operands read before its 16 MFMAs -- and every one of them cost the same 2 050 cycles a term:
    // it is neither the trip to the earlier panels' tiles nor the LDS round trips in front of the matrix instructions.)
    stamp(1);
    for (int i = 0; i < 4; ++i) {                      // accumulator element i of wave wq: row (wq >> 1) 16 + (lane >> 4) + 4 i
        const int r = (wq >> 1) * 16 + (lq >> 4) + 4 * i, c = (wq & 1) * 16 + (lq & 15);
        Dm[r * LS + c] = B[(size_t)(k * BT + r) * ld + k * BT + c] + accD[i];
        if (I != k) La[r * LS + c] = B[(size_t)(I * BT + r) * ld + k * BT + c] + accT[i];
        Wp[r * LS + c] = (r == c) ? 1.0 : 0.0;
    }
    __syncthreads();
    if (WAVE_ELIM) {
        // L_D^-1 of the 32 x 32 block by halving, all of it in ONE wave (no barrier inside: the wave's own LDS accesses are ordered):
        //   W11 = inv chol(D11);  L21 = D21 W11^T;  S = D22 - L21 L21^T;  W22 = inv chol(S);  W21 = -W22 (L21 W11)
        // two 16-step eliminations (wave_inv_chol16) and five 16^3 products on the matrix cores, instead of 32 steps with a
        // workgroup barrier each (measured per panel launch: see DESIGN 4.8).  The block's memory is overwritten as it goes.
        if (wq == 0) {
            double* const D21 = Dm + 16 * LS;            // rows 16.., columns 0..15: D21 -> L21 -> L21 W11
            double* const D22 = Dm + 16 * LS + 16;
            double* const W22 = Wp + 16 * LS + 16;
            const int il = lq & 15, kq = lq >> 4;
            bool bad = wave_inv_chol16(Dm, Wp, rs, lq);
            d4 t = wave_mm16([&](int i, int kk) { return D21[i * LS + kk]; }, [&](int kk, int j) { return Wp[j * LS + kk]; }, lq);
#pragma unroll
            for (int u = 0; u < 4; ++u) D21[(kq + 4 * u) * LS + il] = t[u];
            t = wave_mm16([&](int i, int kk) { return D21[i * LS + kk]; }, [&](int kk, int j) { return D21[j * LS + kk]; }, lq);
#pragma unroll
            for (int u = 0; u < 4; ++u) D22[(kq + 4 * u) * LS + il] -= t[u];
            bad = wave_inv_chol16(D22, W22, rs, lq) || bad;
            t = wave_mm16([&](int i, int kk) { return D21[i * LS + kk]; }, [&](int kk, int j) { return Wp[kk * LS + j]; }, lq);
#pragma unroll
            for (int u = 0; u < 4; ++u) D21[(kq + 4 * u) * LS + il] = t[u];
            t = wave_mm16([&](int i, int kk) { return W22[i * LS + kk]; }, [&](int kk, int j) { return D21[kk * LS + j]; }, lq);
#pragma unroll
            for (int u = 0; u < 4; ++u) Wp[(16 + kq + 4 * u) * LS + il] = -t[u];
            if (lq == 0) sflag = bad ? 1 : 0;
        }
        __syncthreads();
        if (sflag) {                                   // the same block in every workgroup of the panel: all leave
            if (tid == 0 && blockIdx.x == 0) flag[z] = 1;
            return;
        }
    } else
    // [D | I] -> [ . | L_D^-1 ] by elimination with the rows in REGISTERS (round 4): thread (r, cq) holds columns 8 cq .. 8 cq + 7 of
    // row r of the augmented matrix; a step publishes the pivot row and the pivot column through LDS (two alternating buffers:
    // one barrier per step), everything else is eight multiply-adds per thread.  Before, every element of both halves went
    // through LDS in every step (three reads and a write each, 0.45 us a step; 25-36 us of a panel launch were this loop).
    {
        const int er = tid >> 3, cq = tid & 7;
        double a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = cq < 4 ? Dm[er * LS + 8 * cq + i] : ((8 * (cq - 4) + i == er) ? 1.0 : 0.0);
        double* const prow = La;                   // [2][64]   (La holds T only for I != k: parked in registers below)
        double* const pcol = Lb;                   // [2][32]
        double tsave[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) tsave[i] = La[((wq >> 1) * 16 + (lq >> 4) + 4 * i) * LS + (wq & 1) * 16 + (lq & 15)];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < BT; ++j) {
            const int par = j & 1;
            if (er == j) {
#pragma unroll
                for (int i = 0; i < 8; ++i) prow[par * 64 + 8 * cq + i] = a[i];
            }
            if (cq == (j >> 3)) pcol[par * 32 + er] = a[j & 7];
            __syncthreads();
            const double d = pcol[par * 32 + j];
            if (!(d > 0.0) || !(d < 1e300)) {          // the same value in every workgroup of the panel: all leave
                if (tid == 0 && blockIdx.x == 0) flag[z] = 1;
                return;
            }
            if (tid == 0) rs[j] = 1.0 / sqrt(d);
            if (er > j) {
                const double f = pcol[par * 32 + er] * (1.0 / d);
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(-f, prow[par * 64 + 8 * cq + i], a[i]);
            }
        }
        __syncthreads();
        if (cq >= 4) {
            const double sc = rs[er];
#pragma unroll
            for (int i = 0; i < 8; ++i) Wp[er * LS + 8 * (cq - 4) + i] = a[i] * sc;                 // L_D^-1 (lower)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) La[((wq >> 1) * 16 + (lq >> 4) + 4 * i) * LS + (wq & 1) * 16 + (lq & 15)] = tsave[i];
        __syncthreads();
    }
    stamp(2);
    if (I == k) {
        // only the inverse of the diagonal block is ever used again; B[K,K] itself must stay as it is, the other
        // workgroups of this launch are still reading it
        for (int i = 0; i < 4; ++i) {
            const int r = ty + ((i >> 1) << 4), c = tx + ((i & 1) << 4);
            LiBuf[(size_t)k * BT * BT + r * BT + c] = Wp[r * LS + c];
        }
        return;
    }
    __syncthreads();
    const d4 o = mm32_mfma<false, true>(La, Wp, wq, lq);           // T L_D^-T (on the matrix cores too: 3 us on the vector ALU)
    for (int i = 0; i < 4; ++i) {
        const int r = (wq >> 1) * 16 + (lq >> 4) + 4 * i, c = (wq & 1) * 16 + (lq & 15);
        B[(size_t)(I * BT + r) * ld + k * BT + c] = o[i];
    }
    stamp(3);
}

// W = L^-1 by forward substitution on tiles: workgroup (K, cq) owns 8 columns of block column K and walks down
// the block rows; the tiles it wrote are re-read through global memory after a workgroup barrier.
// The terms L_IJ W_JK of a block row are independent of one another: they are taken FOUR tiles of L per pair of barriers (all
// sixteen loads of a thread in flight together; one tile per pair of barriers left the walk at ~1.9 us a term, 300 terms for
// block column 0 at n = 800: 0.58 ms).
constexpr int TI_CHUNK = 4;
__global__ void __launch_bounds__(256) tri_inverse_kernel(int ld, int nbk, const double* __restrict__ Lm,
                                                          const double* __restrict__ LiBuf, double* W,
                                                          const int* __restrict__ flag, size_t mat_stride) {
    __shared__ double Lt[TI_CHUNK][BT * LS], Li[BT * LS], Wt[TI_CHUNK][BT * 8], Acc[BT * 8];
    const int z = blockIdx.z;
    if (flag[z]) return;
    Lm += z * mat_stride;
    W += z * mat_stride;
    LiBuf += (size_t)z * nbk * BT * BT;
    const int K = blockIdx.x, cq = blockIdx.y;
    const int tid = threadIdx.x, t = tid >> 3, u = tid & 7, c = cq * 8 + u;
    for (int I = K; I < nbk; ++I) {
#pragma unroll
        for (int e4 = 0; e4 < BT * BT / 256; ++e4) {
            const int e = tid + 256 * e4;
            Li[(e >> 5) * LS + (e & 31)] = LiBuf[(size_t)I * BT * BT + e];
        }
        double acc = 0.0;
        for (int J0 = K; J0 < I; J0 += TI_CHUNK) {
            const int nj = I - J0 < TI_CHUNK ? I - J0 : TI_CHUNK;
            __syncthreads();
#pragma unroll
            for (int jj = 0; jj < TI_CHUNK; ++jj) {
                if (jj < nj) {
                    const int J = J0 + jj;
#pragma unroll
                    for (int e4 = 0; e4 < BT * BT / 256; ++e4) {
                        const int e = tid + 256 * e4;
                        Lt[jj][(e >> 5) * LS + (e & 31)] = Lm[(size_t)(I * BT + (e >> 5)) * ld + J * BT + (e & 31)];
                    }
                    Wt[jj][t * 8 + u] = W[(size_t)(J * BT + t) * ld + K * BT + c];
                }
            }
            __syncthreads();
            for (int jj = 0; jj < nj; ++jj) {
#pragma unroll 8
                for (int m = 0; m < BT; ++m) acc += Lt[jj][t * LS + m] * Wt[jj][m * 8 + u];
            }
        }
        double val;
        if (I == K) {
            __syncthreads();
            val = Li[t * LS + c];
        } else {
            Acc[t * 8 + u] = acc;
            __syncthreads();
            val = 0.0;
#pragma unroll 8
            for (int m = 0; m < BT; ++m) val -= Li[t * LS + m] * Acc[m * 8 + u];
        }
        W[(size_t)(I * BT + t) * ld + K * BT + c] = val;
        __threadfence_block();
        __syncthreads();
    }
}

// Working copies with leading dimension ld >= n: Bw = B + reg I with a unit diagonal on the ghost rows of the padding
// (so that the factorisation runs through), C0 = A with zero ghosts.                                   apvast.py:24
__global__ void __launch_bounds__(TPB) load_pair_kernel(int n, int ne, int ld, const double* __restrict__ A,
                                                        const double* __restrict__ B, double reg,
                                                        const double* __restrict__ reg_scale, double* __restrict__ C0,
                                                        double* __restrict__ Bw, size_t mat_stride) {
    const int z = blockIdx.z, i = blockIdx.y;
    if (reg_scale != nullptr) reg *= reg_scale[z];           // relative loading: reg ||B||_2 (apvast.py:26-27)
    A += (size_t)z * n * n;
    B += (size_t)z * n * n;
    C0 += z * mat_stride;
    Bw += z * mat_stride;
    for (int j = blockIdx.x * TPB + threadIdx.x; j < ne; j += gridDim.x * TPB) {
        const bool in = i < n && j < n;
        C0[(size_t)i * ld + j] = in ? A[(size_t)i * n + j] : 0.0;
        double b = in ? B[(size_t)i * n + j] : 0.0;
        if (i == j) b = in ? b + reg : 1.0;
        Bw[(size_t)i * ld + j] = b;
    }
}

// X = W^T
__global__ void __launch_bounds__(TPB) transpose_kernel(int n, int ld, const double* __restrict__ W, double* __restrict__ X,
                                                        size_t mat_stride) {
    W += blockIdx.z * mat_stride;
    X += blockIdx.z * mat_stride;
    const int i = blockIdx.y;
    for (int j = blockIdx.x * TPB + threadIdx.x; j < n; j += gridDim.x * TPB) X[(size_t)i * ld + j] = W[(size_t)j * ld + i];
}

// ---- step 2: C = op(A) op(B) on the f64 MFMA: a workgroup owns a 32 x 32 tile of C, the operands pass through LDS in
// 32-deep slices, each of the four waves accumulates one 16 x 16 quarter ------------------------------------------
template <bool TA, bool TB>
__global__ void __launch_bounds__(256) gemm_kernel(int n, int ld, const double* __restrict__ A,
                                                   const double* __restrict__ Bm, double* __restrict__ C,
                                                   size_t mat_stride) {
    __shared__ double sa[BT * LS], sb[BT * LS];          // sa[i][k] = op(A)[row0 + i][k0 + k], sb[k][j] = op(B)[k0 + k][col0 + j]
    const int z = blockIdx.z;
    A += z * mat_stride;
    Bm += z * mat_stride;
    C += z * mat_stride;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int row0 = blockIdx.y * BT, col0 = blockIdx.x * BT;
    d4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < n; k0 += BT) {
        for (int e = tid; e < BT * BT; e += 256) {
            const int r = e >> 5, c = e & 31;            // c runs along the contiguous dimension of the source
            {
                const int i = TA ? c : r, kk = TA ? r : c;                 // source element (r, c) of A's 32 x 32 window
                const int gi = row0 + i, gk = k0 + kk;
                const double v = (gi < n && gk < n) ? (TA ? A[(size_t)gk * ld + gi] : A[(size_t)gi * ld + gk]) : 0.0;
                sa[i * LS + kk] = v;
            }
            {
                const int kk = TB ? c : r, j = TB ? r : c;
                const int gk = k0 + kk, gj = col0 + j;
                const double v = (gk < n && gj < n) ? (TB ? Bm[(size_t)gj * ld + gk] : Bm[(size_t)gk * ld + gj]) : 0.0;
                sb[kk * LS + j] = v;
            }
        }
        __syncthreads();
        const d4 part = mm32_mfma<false>(sa, sb, w, lane);
        for (int t = 0; t < 4; ++t) acc[t] += part[t];
        __syncthreads();
    }
    const int orow = row0 + (w >> 1) * 16 + (lane >> 4), ocol = col0 + (w & 1) * 16 + (lane & 15);
    for (int t = 0; t < 4; ++t)
        if (orow + 4 * t < n && ocol < n) C[(size_t)(orow + 4 * t) * ld + ocol] = acc[t];
}

// ---- round 4: a general product for the factor-and-whiten stage --------------------------------------------------------------
// C = alpha op(A) op(B), 64 x 64 tile per workgroup, each of the four waves a 32 x 32 quarter (2 x 2 MFMA tiles), operands
// straight from global memory in the MFMA's own layout (no LDS: the two waves that share an operand strip hit the same lines
// of the CU's L1), K in chunks of 16 with the next chunk's loads in flight during this chunk's sixteen MFMAs.
//   A operand, element (i, k): A[i lda + k]  (one 32-byte load per lane and chunk: lane (r, kq) holds k0 + 4 kq .. + 3; MFMA j of
//                                             the chunk contracts k0 + 4 kq + j -- any order of the contraction index serves)
//   B operand, element (k, j): TB ? B[j ldb + k] (the same 32-byte pattern) : B[k ldb + j] (four 8-byte loads, coalesced over j)
// Two levels of batch: blockIdx.z = z * n2 + p, pointer = base + z * s?1 + p * s?2 (p: the pairs of a level of tri_invert).
// klim 1: A is lower triangular, k stops at the tile's last row; klim 2: op(B) = W^T with W lower triangular, k stops at the tile's
// last column.  lower_only: tiles strictly above the diagonal are not computed (the caller mirrors).
struct Gemm64 {
    int M, N, K;
    const double* A; int lda; size_t sA1, sA2;
    const double* B; int ldb; size_t sB1, sB2;
    double* C; int ldc; size_t sC1, sC2;
    double alpha;
    int n2, klim, lower_only;
    const int* flag;        // per matrix z: non-zero = the factorisation failed, leave
};

template <bool TB>
__global__ void __launch_bounds__(256) gemm64_kernel(const Gemm64 g) {
    const int zz = blockIdx.z, z = zz / g.n2, pp = zz % g.n2;
    if (g.flag && g.flag[z]) return;
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    if (g.lower_only && col0 > row0) return;
    const double* A = g.A + z * g.sA1 + pp * g.sA2;
    const double* B = g.B + z * g.sB1 + pp * g.sB2;
    double* C = g.C + z * g.sC1 + pp * g.sC2;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, kq = lane >> 4;
    const int wr = row0 + (w >> 1) * 32, wc = col0 + (w & 1) * 32;
    int kend = g.K;
    if (g.klim == 1) kend = min(kend, row0 + 64);
    if (g.klim == 2) kend = min(kend, col0 + 64);
    kend = (kend + 15) & ~15;
    if (kend > g.K) kend = g.K;             // K is a multiple of 16 at every call site
    // rows / columns beyond M / N are clamped for the loads and masked at the store
    const int ar0 = min(wr + il, g.M - 1), ar1 = min(wr + 16 + il, g.M - 1);
    const int bc0 = min(wc + il, g.N - 1), bc1 = min(wc + 16 + il, g.N - 1);
    const double* a0 = A + (size_t)ar0 * g.lda + 4 * kq;
    const double* a1 = A + (size_t)ar1 * g.lda + 4 * kq;
    d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = d4{0, 0, 0, 0};
    d4 av[2], bv[2], nav[2], nbv[2];
    auto fetch = [&](int k0, d4 (&ta)[2], d4 (&tb)[2]) {
        ta[0] = *reinterpret_cast<const d4*>(a0 + k0);
        ta[1] = *reinterpret_cast<const d4*>(a1 + k0);
        if (TB) {
            tb[0] = *reinterpret_cast<const d4*>(B + (size_t)bc0 * g.ldb + k0 + 4 * kq);
            tb[1] = *reinterpret_cast<const d4*>(B + (size_t)bc1 * g.ldb + k0 + 4 * kq);
        } else {
            const double* br = B + (size_t)(k0 + 4 * kq) * g.ldb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tb[0][j] = br[(size_t)j * g.ldb + bc0];
                tb[1][j] = br[(size_t)j * g.ldb + bc1];
            }
        }
    };
    if (kend > 0) fetch(0, av, bv);
    for (int k0 = 0; k0 < kend; k0 += 16) {
        if (k0 + 16 < kend) fetch(k0 + 16, nav, nbv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0][j], bv[0][j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0][j], bv[1][j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1][j], bv[0][j], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1][j], bv[1][j], acc[1][1], 0, 0, 0);
        }
        av[0] = nav[0]; av[1] = nav[1]; bv[0] = nbv[0]; bv[1] = nbv[1];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int r = wr + 16 * i + kq + 4 * t, c = wc + 16 * j + il;
                if (r < g.M && c < g.N) C[(size_t)r * g.ldc + c] = g.alpha * acc[i][j][t];
            }
}

// C[j][i] = C[i][j] for j < i (the whitened matrix was formed on and below the diagonal tiles only)
__global__ void __launch_bounds__(TPB) mirror_lower_kernel(int n, int ld, double* __restrict__ C, size_t mat_stride, const int* __restrict__ flag) {
    if (flag[blockIdx.z]) return;
    C += blockIdx.z * mat_stride;
    const int i = blockIdx.y;
    for (int j = blockIdx.x * TPB + threadIdx.x; j < i; j += gridDim.x * TPB) C[(size_t)j * ld + i] = C[(size_t)i * ld + j];
}

// W's diagonal 32 x 32 blocks = the inverses of L's diagonal blocks that the panel kernel left in LiBuf
__global__ void __launch_bounds__(256) diag_inverse_scatter_kernel(int ld, int nbk, const double* __restrict__ LiBuf, double* __restrict__ W,
                                                                   size_t mat_stride, const int* __restrict__ flag) {
    const int z = blockIdx.z, k = blockIdx.x;
    if (flag[z]) return;
    const double* src = LiBuf + ((size_t)z * nbk + k) * BT * BT;
    double* dst = W + z * mat_stride + (size_t)k * BT * ld + k * BT;
    for (int e = threadIdx.x; e < BT * BT; e += 256) dst[(size_t)(e >> 5) * ld + (e & 31)] = src[e];
}

__global__ void __launch_bounds__(TPB) symmetrise_kernel(int n, int ld, double* __restrict__ C, size_t mat_stride) {
    C += blockIdx.z * mat_stride;
    const int i = blockIdx.y;
    for (int j = blockIdx.x * TPB + threadIdx.x; j < i; j += gridDim.x * TPB) {
        const double m = 0.5 * (C[(size_t)i * ld + j] + C[(size_t)j * ld + i]);
        C[(size_t)i * ld + j] = m;
        C[(size_t)j * ld + i] = m;
    }
}

// ---- step 3 ---------------------------------------------------------------------------------------
__device__ __forceinline__ void rr_pair(int ne, int r, int a, int& p, int& q) {
    const int m1 = ne - 1;
    int u, v;
    if (a == 0) { u = m1; v = r; } else { u = (r + a) % m1; v = (r - a + m1) % m1; }
    p = u < v ? u : v;
    q = u < v ? v : u;
}

// 1/sqrt(x) to full double precision: v_rsq_f64 (5e-8 on gfx950) and one third-order correction
__device__ __forceinline__ double rsq_f64(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
}

// Jacobi rotation for [[alpha, beta], [beta, gamma]], branch-free.  The tangent is evaluated in float (the
// double-precision divide and square root cost ~10x more on the serial path of every inner round); (c, s) are
// then formed in double from that tangent, so the rotation is orthogonal to 1e-16 but leaves ~1e-7 |beta| in the
// pivot instead of an exact zero -- the caller keeps the computed pivot, and the sweeps still converge
// quadratically.  With tau = (gamma - alpha) / (2 beta): t = sign(tau) / (|tau| + sqrt(1 + tau^2)), written on
// rho = min(|d|, |2 beta|) / max(|d|, |2 beta|) so that nothing overflows.
__device__ __forceinline__ void sym_rotation(double alpha, double gamma, double beta, double& c, double& s) {
    const double b2 = beta * beta;
    const bool rotate = b2 > 1e-290 && b2 > 1e-60 * (alpha * alpha + gamma * gamma);
    const double d = gamma - alpha, tb = 2.0 * beta;
    // only the ratio matters: bring the larger of the two to [0.5, 1) before leaving double range; then
    // |t| = |2 beta| / (|d| + sqrt(d^2 + 4 beta^2)) needs no case distinction (round 3: the inner round's critical path is this
    // function, evaluated by sixteen lanes while everybody else waits; 45 -> 27 instructions)
    const int ex = __builtin_amdgcn_frexp_exp(fabs(d) >= fabs(tb) ? d : tb);
    const float fd = (float)__builtin_ldexp(d, -ex), fb = (float)__builtin_ldexp(tb, -ex);
    const float s2 = __builtin_fmaf(fd, fd, fb * fb);                // in [0.25, 2)
    const float hyp = s2 * __builtin_amdgcn_rsqf(s2);
    const float mag = fabsf(fb) * __builtin_amdgcn_rcpf(fabsf(fd) + hyp);
    const float t = __builtin_copysignf(mag, __builtin_copysignf(1.0f, fd) * fb);
    const double td = (double)t;
    const double cc = rsq_f64(__builtin_fma(td, td, 1.0));
    c = rotate ? cc : 1.0;
    s = rotate ? td * cc : 0.0;
}

__device__ __forceinline__ int tile_index(int I, int J, int t) { return t < BH ? I * BH + t : J * BH + t - BH; }

// One block round.  Tile (P, Q), P <= Q, of the pair grid: rows = the two 16-blocks of pair P, columns = those of
// pair Q.  The workgroup runs the inner sweep on the diagonal tiles (P, P) and (Q, Q) side by side, one per half of
// its 512 threads: thread (a, b) of a half owns the 2 x 2 block rows {p_a, q_a} x columns {p_b, q_b} of its tile and
// rows {2a, 2a+1} of the accumulated rotation V; thread (0, b) derives the rotation of pair b and publishes it (the
// inner rounds are bound by instruction issue, so the rotations are worked out in one wave per tile only).
// Then V_P^T C[P,Q] V_Q (and its transpose) goes to the other global buffer and the two tiles of X this workgroup
// owns are rotated in place.
// `full`: the inner sweep visits all 496 pairs of the 32 indices (first round of a sweep: that is where the pairs
// inside one 16-block are annihilated); otherwise only the 256 pairs across the two blocks, in 16 rounds.
//
// MODE 0: all of the above in one launch (np (np + 1) / 2 workgroups; rounds 1-2, now the A/B switch APV_LARGE_SPLIT=0).  The
// redundant inner sweeps cost real time once a launch fills the chip (n = 800: 25 pair problems solved by 650 workgroups), and
// even below that the off-diagonal workgroups are better off not sweeping.  Round 3 splits the round: MODE 1, np workgroups,
// solves the pair problems (the diagonal tiles: inner sweep, rotated tile, its X tile) and leaves the rotations V_P in Vbuf;
// MODE 2, np (np - 1) / 2 workgroups, takes V_P, V_Q from there and applies them to its off-diagonal tile and its two X tiles.
template <int MODE>
__device__ __forceinline__ void block_jacobi_round_body(double* sm, double2 (*rot)[BH], int blk, int ld, int nb, int round, int full,
                                                        const double* __restrict__ Cin, double* __restrict__ Cout,
                                                        double* __restrict__ X, double* __restrict__ off, size_t mat_stride,
                                                        double* __restrict__ Vbuf, const double* __restrict__ Dbuf) {
    constexpr int TS = BT * LS;
    double *SP = sm, *SQ = sm + TS, *VP = sm + 2 * TS, *VQ = sm + 3 * TS, *T = sm + 4 * TS, *U = sm + 5 * TS;
    const int z = blockIdx.z;
    Cin += z * mat_stride;
    Cout += z * mat_stride;
    X += z * mat_stride;
    const int np = nb / 2;
    int P = 0, rem = blk;
    // MODE 5 (look-ahead, see la_solve_kernel): the update half of a round for ALL tiles -- diagonal workgroups copy the rotated
    // tile from Dbuf and rotate their X tile, the others are MODE 2
    if (MODE == 1) {
        P = blk;
        rem = 0;
    } else {
        const int skip = (MODE == 2) ? 1 : 0;                 // MODE 2 enumerates the tiles above the diagonal only
        while (rem >= np - P - skip) {
            rem -= np - P - skip;
            ++P;
        }
        rem += skip;
    }
    const int Q = P + rem;
    const bool diag = P == Q;
    int IP, JP, IQ, JQ;
    rr_pair(nb, round, P, IP, JP);
    rr_pair(nb, round, Q, IQ, JQ);
    const int tid = threadIdx.x, half = tid >> 8, a = (tid >> 4) & 15, b = tid & 15;
    double* const Vz = Vbuf + (size_t)z * np * TS;
    #pragma unroll
    for (int e2 = 0; e2 < BT * BT / 512; ++e2) {
        const int e = tid + 512 * e2;             // (all of a thread's loads go out before the first is used)
        const int t = e >> 5, u = e & 31;
        const size_t gr = (size_t)tile_index(IP, JP, t) * ld;
        if (MODE == 5 && diag) {                              // look-ahead: the pair solve has left the rotated tile and V_P behind
            SP[t * LS + u] = Dbuf[((size_t)z * np + P) * TS + t * LS + u];
            VP[t * LS + u] = Vz[(size_t)P * TS + t * LS + u];
        } else if (MODE != 2 && MODE != 5) {
            SP[t * LS + u] = Cin[gr + tile_index(IP, JP, u)];
            VP[t * LS + u] = (t == u) ? 1.0 : 0.0;
        } else {
            VP[t * LS + u] = Vz[(size_t)P * TS + t * LS + u];
            VQ[t * LS + u] = Vz[(size_t)Q * TS + t * LS + u];
            T[t * LS + u] = Cin[gr + tile_index(IQ, JQ, u)];
        }
        if (MODE == 0 && !diag) {
            SQ[t * LS + u] = Cin[(size_t)tile_index(IQ, JQ, t) * ld + tile_index(IQ, JQ, u)];
            T[t * LS + u] = Cin[gr + tile_index(IQ, JQ, u)];
            VQ[t * LS + u] = (t == u) ? 1.0 : 0.0;
        }
    }
    __syncthreads();
    // ---- inner sweep -------------------------------------------------------------------------------
    const bool active = half == 0 || !diag;
    double* const S = half ? SQ : SP;
    double* const V = half ? VQ : VP;
    const int inner_rounds = (MODE == 2 || MODE == 5) ? 0 : (full ? BT - 1 : BH);
    // round-robin positions of pair a / pair b, advanced incrementally (rr_pair without the modulo)
    int ua = (a == 0) ? BT - 1 : a, va = (a == 0) ? 0 : BT - 1 - a;
    int ub = (b == 0) ? BT - 1 : b, vb = (b == 0) ? 0 : BT - 1 - b;
    double offacc = 0.0;
    for (int t = 0; t < inner_rounds; ++t) {
        int pa, qa, pb, qb;
        if (full) {
            pa = min(ua, va); qa = max(ua, va);
            pb = min(ub, vb); qb = max(ub, vb);
            if (a != 0) ua = (ua + 1 == BT - 1) ? 0 : ua + 1;
            va = (va + 1 == BT - 1) ? 0 : va + 1;
            if (b != 0) ub = (ub + 1 == BT - 1) ? 0 : ub + 1;
            vb = (vb + 1 == BT - 1) ? 0 : vb + 1;
        } else {
            pa = a;
            qa = BH + ((a + t) & 15);
            pb = b;
            qb = BH + ((b + t) & 15);
        }
        if (active && a == 0) {                  // the 16 leading lanes of one wave per half; the other waves skip
            const double beta = S[pb * LS + qb];
            double c, s;
            sym_rotation(S[pb * LS + pb], S[qb * LS + qb], beta, c, s);
            rot[half][b] = make_double2(c, s);
            if (half == 0) offacc += beta * beta;
        }
        __syncthreads();
        if (active) {
            const double2 ra = rot[half][a], rb = rot[half][b];
            const double ca = ra.x, sa = ra.y, cb = rb.x, sb = rb.y;
            const double xpp = S[pa * LS + pb], xpq = S[pa * LS + qb], xqp = S[qa * LS + pb], xqq = S[qa * LS + qb];
            const int r0 = 2 * a, r1 = 2 * a + 1;
            const double v0p = V[r0 * LS + pb], v0q = V[r0 * LS + qb], v1p = V[r1 * LS + pb], v1q = V[r1 * LS + qb];
            // columns: [x_p, x_q] J_b, J = [[c, s], [-s, c]]; rows: J_a^T [y_p; y_q]
            const double ypp = cb * xpp - sb * xpq, ypq = sb * xpp + cb * xpq;
            const double yqp = cb * xqp - sb * xqq, yqq = sb * xqp + cb * xqq;
            const double zpq = ca * ypq - sa * yqq;
            S[pa * LS + pb] = ca * ypp - sa * yqp;
            S[pa * LS + qb] = zpq;
            S[qa * LS + pb] = (a == b) ? zpq : sa * ypp + ca * yqp;       // the pivot keeps its (tiny) computed value
            S[qa * LS + qb] = sa * ypq + ca * yqq;
            V[r0 * LS + pb] = cb * v0p - sb * v0q;
            V[r0 * LS + qb] = sb * v0p + cb * v0q;
            V[r1 * LS + pb] = cb * v1p - sb * v1q;
            V[r1 * LS + qb] = sb * v1p + cb * v1q;
        }
        __syncthreads();
    }
    // ---- outer update: 32 x 32 x 32 products on the f64 MFMA, one 16 x 16 output tile per wave ------------
    const int w = (tid >> 6) & 3, lane = tid & 63;
    const int orow = (w >> 1) * 16 + (lane >> 4), ocol = (w & 1) * 16 + (lane & 15);     // + 4 t on the row
    if (diag) {
        if (tid < 16 && offacc != 0.0) atomicAdd(off + z, offacc);
        #pragma unroll
        for (int e2 = 0; e2 < BT * BT / 512; ++e2) {
            const int e = tid + 512 * e2;             // (all of a thread's loads go out before the first is used)
            const int t = e >> 5, u = e & 31;
            Cout[(size_t)tile_index(IP, JP, t) * ld + tile_index(IP, JP, u)] = SP[t * LS + u];
            T[t * LS + u] = X[(size_t)(P * BT + t) * ld + tile_index(IP, JP, u)];
            if (MODE == 1) Vz[(size_t)P * TS + t * LS + u] = VP[t * LS + u];
        }
        __syncthreads();
        if (half) return;
        const d4 o = mm32_mfma<false>(T, VP, w, lane);
        for (int t = 0; t < 4; ++t) X[(size_t)(P * BT + orow + 4 * t) * ld + tile_index(IP, JP, ocol)] = o[t];
        return;
    }
    // half 0: U = C[P,Q] V_Q, then V_P^T U; half 1 meanwhile rotates the two X tiles (staged in the buffers of the
    // rotated diagonal tiles, which are not needed in an off-diagonal workgroup)
    if (half == 0) {
        const d4 o = mm32_mfma<false>(T, VQ, w, lane);
        for (int t = 0; t < 4; ++t) U[(orow + 4 * t) * LS + ocol] = o[t];
    }
    #pragma unroll
    for (int e2 = 0; e2 < BT * BT / 512; ++e2) {
        const int e = tid + 512 * e2;             // (all of a thread's loads go out before the first is used)
        const int t = e >> 5, u = e & 31;
        SP[t * LS + u] = X[(size_t)(Q * BT + t) * ld + tile_index(IP, JP, u)];
        SQ[t * LS + u] = X[(size_t)(P * BT + t) * ld + tile_index(IQ, JQ, u)];
    }
    __syncthreads();
    if (half == 0) {
        const d4 o = mm32_mfma<true>(VP, U, w, lane);
        for (int t = 0; t < 4; ++t) {
            Cout[(size_t)tile_index(IP, JP, orow + 4 * t) * ld + tile_index(IQ, JQ, ocol)] = o[t];
            T[(orow + 4 * t) * LS + ocol] = o[t];                   // every thread is past its reads of T
        }
    } else {
        const d4 x1 = mm32_mfma<false>(SP, VP, w, lane), x2 = mm32_mfma<false>(SQ, VQ, w, lane);
        for (int t = 0; t < 4; ++t) {
            X[(size_t)(Q * BT + orow + 4 * t) * ld + tile_index(IP, JP, ocol)] = x1[t];
            X[(size_t)(P * BT + orow + 4 * t) * ld + tile_index(IQ, JQ, ocol)] = x2[t];
        }
    }
    __syncthreads();
    #pragma unroll
    for (int e2 = 0; e2 < BT * BT / 512; ++e2) {                     // the mirrored tile, coalesced
        const int e = tid + 512 * e2;             // (all of a thread's loads go out before the first is used)
        const int t = e >> 5, u = e & 31;
        Cout[(size_t)tile_index(IQ, JQ, t) * ld + tile_index(IP, JP, u)] = T[u * LS + t];
    }
}

__global__ void __launch_bounds__(TPB) frob2_kernel(int n, int ld, const double* __restrict__ C, double* __restrict__ out,
                                                    size_t mat_stride) {
    C += blockIdx.z * mat_stride;
    __shared__ double red[TPB];
    double s = 0.0;
    for (int idx = blockIdx.x * TPB + threadIdx.x; idx < n * n; idx += gridDim.x * TPB) {
        const double v = C[(size_t)(idx / n) * ld + (idx % n)];
        s += v * v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = TPB / 2; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out + blockIdx.z, red[0]);
}

// out[0..count) = 0.  A kernel, not hipMemsetAsync, so that the captured two-sweep hipGraph holds kernel nodes only.  Round 1
// saw rocprofv3 --kernel-trace segfault at the first replay of this graph and suspected its memset node; in round 2 the crash
// did not reproduce with or without that node (profiles/r02/rocprof_graph.md: five command lines) -- the likelier causes, a
// per-thread flag read and an unordered upload, were fixed meanwhile.  The kernel node stays: it traces like any other launch.
__global__ void __launch_bounds__(64) zero_f64_kernel(int count, double* __restrict__ out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < count) out[i] = 0.0;
}

// ---- step 4 ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) rank_kernel(int n, int ld, const double* __restrict__ C, double* __restrict__ lam,
                                                   int* __restrict__ order, size_t mat_stride, size_t vec_stride) {
    // the diagonal goes through LDS once (n <= 2048): read from global memory inside the counting loop, every thread walked n
    // strided words one dependent load after the other (180 us at n = 800)
    __shared__ double sdiag[2048];
    C += blockIdx.z * mat_stride;
    lam += blockIdx.z * (size_t)n;              // dense [batch][n]
    order += blockIdx.z * vec_stride;
    for (int j = threadIdx.x; j < n; j += TPB) sdiag[j] = C[(size_t)j * ld + j];
    __syncthreads();
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const double li = sdiag[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
        const double lj = sdiag[j];
        rank += (lj > li) || (lj == li && j < i);
    }
    order[rank] = i;
    lam[rank] = li;
}

// U[i][c] = X[i][order[c]]
__global__ void __launch_bounds__(TPB) gather_cols_kernel(int n, int ld, const double* __restrict__ X,
                                                          const int* __restrict__ order, double* __restrict__ U,
                                                          size_t mat_stride, size_t vec_stride, size_t out_stride) {
    X += blockIdx.z * mat_stride;
    order += blockIdx.z * vec_stride;
    U += blockIdx.z * out_stride;
    const int i = blockIdx.y;
    for (int c = blockIdx.x * TPB + threadIdx.x; c < n; c += gridDim.x * TPB) U[(size_t)i * n + c] = X[(size_t)i * ld + order[c]];
}

// ---- step 5 ---------------------------------------------------------------------------------------
// coef[c] = (u_c . r) / (lam_c + mu) on the sorted eigenvectors U (n x n, dense)
__global__ void __launch_bounds__(TPB) coef_kernel(int n, const double* __restrict__ U, const double* __restrict__ lam,
                                                   const double* __restrict__ r, double mu, double* __restrict__ coef,
                                                   size_t out_stride, size_t vec_stride) {
    // 32 columns per workgroup, the rows dealt to eight groups of threads (one thread per column walked all n rows: 196 us at
    // n = 800); the eight partial sums meet in LDS in a fixed order
    __shared__ double part[8][32];
    U += blockIdx.z * out_stride;
    lam += blockIdx.z * (size_t)n;              // dense [batch][n]
    r += blockIdx.z * (size_t)n;
    coef += blockIdx.z * vec_stride;
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s = 0.0;
    if (c < n)
        for (int i = g; i < n; i += 8) s += U[(size_t)i * n + c] * r[i];
    part[g][cl] = s;
    __syncthreads();
    if (g == 0 && c < n) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += part[q][cl];
        coef[c] = t / (lam[c] + mu);
    }
}

// w[q][i] = sum_{c < ranks[q]} coef[c] U[i][c] for the ascending rank list (ranks == nullptr: every rank 1..V, the
// reference keeps them all, apvast.py:406-414; a list is apVast.m:527-549)
__global__ void __launch_bounds__(TPB) vast_prefix_kernel(int n, int V, const int* __restrict__ ranks,
                                                          const double* __restrict__ U, const double* __restrict__ coef,
                                                          double* __restrict__ w, size_t out_stride, size_t vec_stride,
                                                          size_t w_stride) {
    U += blockIdx.z * out_stride;
    coef += blockIdx.z * vec_stride;
    w += blockIdx.z * w_stride;
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
    int c = 0;
    for (int q = 0; q < V; ++q) {
        const int upto = ranks ? ranks[q] : q + 1;
        for (; c < upto && c < n; ++c) acc += coef[c] * U[(size_t)i * n + c];
        w[(size_t)q * n + i] = acc;
    }
}

}  // namespace

// Workspace + captured two-sweep graph, cached on the handle (sizes rarely change between hops).
template <int MODE>
__global__ void __launch_bounds__(512) block_jacobi_round_kernel(int ld, int nb, int round, int full,
                                                                 const double* __restrict__ Cin, double* __restrict__ Cout,
                                                                 double* __restrict__ X, double* __restrict__ off,
                                                                 size_t mat_stride, double* __restrict__ Vbuf,
                                                                 const double* __restrict__ Dbuf) {
    __shared__ double sm[6 * BT * LS];
    __shared__ double2 rot[2][BH];
    block_jacobi_round_body<MODE>(sm, rot, blockIdx.x, ld, nb, round, full, Cin, Cout, X, off, mat_stride, Vbuf, Dbuf);
}

// ---- look-ahead (round 3) -----------------------------------------------------------------------------------
// The pair problems of round r + 1 do not need the whole matrix after round r: the diagonal tile of the pair (I', J') is made of
// the blocks (I', I'), (J', J') -- sub-blocks of round r's ROTATED diagonal tiles, which the pair solves of round r leave in Dbuf --
// and (I', J'), a sub-block of V_a^T C_r[tile (a, b)] V_b for the two round-r pairs a, b that held I' and J' (never the same pair
// two rounds running).  la_solve_kernel forms that one tile itself (two 32^3 products) and solves, so that the update of round
// r (la: block_jacobi_round_kernel<5>, the whole chip) runs BESIDE the pair solves of round r + 1 (np workgroups, one long
// dependent chain each) on a second stream: a round costs the longer of the two instead of their sum.

// pair index and position (0: first, 1: second block) of block A in round r of the tournament
__device__ __forceinline__ void rr_locate(int ne, int r, int A, int& pair, int& half) {
    const int m1 = ne - 1;
    int a = 0;
    if (A != m1 && A != r) {
        a = (A - r + m1) % m1;
        if (a >= ne / 2) a = m1 - a;
    }
    int p, q;
    rr_pair(ne, r, a, p, q);
    pair = a;
    half = (A == p) ? 0 : 1;
}

// FIRST: the matrix Csrc is complete (first round of a graph launch): the tile is read as it stands.  Otherwise Csrc is the
// matrix BEFORE round `prev_round`, Vprev / Dprev that round's rotations and rotated diagonal tiles.
template <bool FIRST>
__device__ __forceinline__ void la_solve_body(double* sm, double2 (*rot)[BH], int blk, int ld, int nb, int round, int prev_round, int full,
                                              const double* __restrict__ Csrc, const double* __restrict__ Vprev,
                                              const double* __restrict__ Dprev, double* __restrict__ Vcur,
                                              double* __restrict__ Dcur, double* __restrict__ off, size_t mat_stride) {
    constexpr int TS = BT * LS;
    double *SP = sm, *VP = sm + TS, *T = sm + 2 * TS, *U = sm + 3 * TS, *VA = sm + 4 * TS, *VB = sm + 5 * TS;
    const int z = blockIdx.z, np = nb / 2, P = blk;
    __builtin_amdgcn_s_setprio(3);          // the launch lasts as long as this chain: its waves go first beside the update's
    Csrc += z * mat_stride;
    int IP, JP;
    rr_pair(nb, round, P, IP, JP);
    const int tid = threadIdx.x, half = tid >> 8, a = (tid >> 4) & 15, b = tid & 15;
    const int w = (tid >> 6) & 3, lane = tid & 63;
    const int orow = (w >> 1) * 16 + (lane >> 4), ocol = (w & 1) * 16 + (lane & 15);     // + 4 t on the row
    if (FIRST) {
        #pragma unroll
        for (int e2 = 0; e2 < BT * BT / 512; ++e2) {
            const int e = tid + 512 * e2;             // (all of a thread's loads go out before the first is used)
            const int t = e >> 5, u = e & 31;
            SP[t * LS + u] = Csrc[(size_t)tile_index(IP, JP, t) * ld + tile_index(IP, JP, u)];
            VP[t * LS + u] = (t == u) ? 1.0 : 0.0;
        }
        __syncthreads();
    } else {
        int pa, ha, pb, hb, Ia, Ja, Ib, Jb;
        rr_locate(nb, prev_round, IP, pa, ha);
        rr_locate(nb, prev_round, JP, pb, hb);
        rr_pair(nb, prev_round, pa, Ia, Ja);
        rr_pair(nb, prev_round, pb, Ib, Jb);
        const double* Va = Vprev + ((size_t)z * np + pa) * TS;
        const double* Vb = Vprev + ((size_t)z * np + pb) * TS;
        const double* Da = Dprev + ((size_t)z * np + pa) * TS;
        const double* Db = Dprev + ((size_t)z * np + pb) * TS;
        #pragma unroll
        for (int e2 = 0; e2 < BT * BT / 512; ++e2) {
            const int e = tid + 512 * e2;             // (all of a thread's loads go out before the first is used)
            const int t = e >> 5, u = e & 31;
            T[t * LS + u] = Csrc[(size_t)tile_index(Ia, Ja, t) * ld + tile_index(Ib, Jb, u)];
            VA[t * LS + u] = Va[t * LS + u];
            VB[t * LS + u] = Vb[t * LS + u];
            VP[t * LS + u] = (t == u) ? 1.0 : 0.0;
            if (t < BH && u < BH) {                            // the two diagonal blocks, from the rotated tiles of the last round
                SP[t * LS + u] = Da[(ha * BH + t) * LS + ha * BH + u];
                SP[(BH + t) * LS + BH + u] = Db[(hb * BH + t) * LS + hb * BH + u];
            }
        }
        __syncthreads();
        if (half == 0) {
            const d4 o = mm32_mfma<false>(T, VB, w, lane);
            for (int t = 0; t < 4; ++t) U[(orow + 4 * t) * LS + ocol] = o[t];
        }
        __syncthreads();
        if (half == 0) {
            const d4 o = mm32_mfma<true>(VA, U, w, lane);       // V_a^T (C V_b)
            for (int t = 0; t < 4; ++t) T[(orow + 4 * t) * LS + ocol] = o[t];
        }
        __syncthreads();
        for (int e = tid; e < BH * BH; e += 512) {
            const int t = e >> 4, u = e & 15;
            const double v = T[(ha * BH + t) * LS + hb * BH + u];
            SP[t * LS + BH + u] = v;
            SP[(BH + u) * LS + t] = v;
        }
        __syncthreads();
    }
    // ---- inner sweep (the first half of the workgroup; see block_jacobi_round_kernel) ----
    // the first half of the workgroup rotates the tile, the second the accumulated rotation V (both take the round's rotations
    // from `rot`): half the reads and flops per thread in the part of an inner round that all threads share
    const int inner_rounds = full ? BT - 1 : BH;
    int ua = (a == 0) ? BT - 1 : a, va = (a == 0) ? 0 : BT - 1 - a;
    int ub = (b == 0) ? BT - 1 : b, vb = (b == 0) ? 0 : BT - 1 - b;
    double offacc = 0.0;
    for (int t = 0; t < inner_rounds; ++t) {
        int pa, qa, pb, qb;
        if (full) {
            pa = min(ua, va); qa = max(ua, va);
            pb = min(ub, vb); qb = max(ub, vb);
            if (a != 0) ua = (ua + 1 == BT - 1) ? 0 : ua + 1;
            va = (va + 1 == BT - 1) ? 0 : va + 1;
            if (b != 0) ub = (ub + 1 == BT - 1) ? 0 : ub + 1;
            vb = (vb + 1 == BT - 1) ? 0 : vb + 1;
        } else {
            pa = a;
            qa = BH + ((a + t) & 15);
            pb = b;
            qb = BH + ((b + t) & 15);
        }
        if (half == 0 && a == 0) {
            const double beta = SP[pb * LS + qb];
            double c, s;
            sym_rotation(SP[pb * LS + pb], SP[qb * LS + qb], beta, c, s);
            rot[0][b] = make_double2(c, s);
            offacc += beta * beta;
        }
        __syncthreads();
        if (half == 0) {
            const double2 ra = rot[0][a], rb = rot[0][b];
            const double ca = ra.x, sa = ra.y, cb = rb.x, sb = rb.y;
            const double xpp = SP[pa * LS + pb], xpq = SP[pa * LS + qb], xqp = SP[qa * LS + pb], xqq = SP[qa * LS + qb];
            const double ypp = cb * xpp - sb * xpq, ypq = sb * xpp + cb * xpq;
            const double yqp = cb * xqp - sb * xqq, yqq = sb * xqp + cb * xqq;
            const double zpq = ca * ypq - sa * yqq;
            SP[pa * LS + pb] = ca * ypp - sa * yqp;
            SP[pa * LS + qb] = zpq;
            SP[qa * LS + pb] = (a == b) ? zpq : sa * ypp + ca * yqp;
            SP[qa * LS + qb] = sa * ypq + ca * yqq;
        } else {
            const double2 rb = rot[0][b];
            const double cb = rb.x, sb = rb.y;
            const int r0 = 2 * a, r1 = 2 * a + 1;
            const double v0p = VP[r0 * LS + pb], v0q = VP[r0 * LS + qb], v1p = VP[r1 * LS + pb], v1q = VP[r1 * LS + qb];
            VP[r0 * LS + pb] = cb * v0p - sb * v0q;
            VP[r0 * LS + qb] = sb * v0p + cb * v0q;
            VP[r1 * LS + pb] = cb * v1p - sb * v1q;
            VP[r1 * LS + qb] = sb * v1p + cb * v1q;
        }
        __syncthreads();
    }
    if (tid < 16 && offacc != 0.0) atomicAdd(off + z, offacc);
    double* Dc = Dcur + ((size_t)z * np + P) * TS;
    double* Vc = Vcur + ((size_t)z * np + P) * TS;
    #pragma unroll
    for (int e2 = 0; e2 < BT * BT / 512; ++e2) {
        const int e = tid + 512 * e2;             // (all of a thread's loads go out before the first is used)
        const int t = e >> 5, u = e & 31;
        Dc[t * LS + u] = SP[t * LS + u];
        Vc[t * LS + u] = VP[t * LS + u];
    }
}

template <bool FIRST>
__global__ void __launch_bounds__(512) la_solve_kernel(int ld, int nb, int round, int prev_round, int full,
                                                       const double* __restrict__ Csrc, const double* __restrict__ Vprev,
                                                       const double* __restrict__ Dprev, double* __restrict__ Vcur,
                                                       double* __restrict__ Dcur, double* __restrict__ off, size_t mat_stride) {
    __shared__ double sm[6 * BT * LS];
    __shared__ double2 rot[2][BH];
    la_solve_body<FIRST>(sm, rot, blockIdx.x, ld, nb, round, prev_round, full, Csrc, Vprev, Dprev, Vcur, Dcur, off, mat_stride);
}

// One launch = the update of round r (workgroups np ..) AND the pair solves of round r + 1 (workgroups 0 .. np - 1: dispatched
// first, they are the long chain).  Both read the matrix before round r and round r's rotations; the solves write only the next
// round's V / D buffers, the updates only the next matrix and X: no dependency inside the launch, and consecutive launches are
// ordered by the stream.  (A first version ran the two on separate streams joined by events, in the captured graph and out of
// it: the cross-stream edges cost ~10 us a round, most of what the overlap buys.)
struct LaRound {
    int round, full;               // the round being updated / whether the NEXT round's inner sweep is a full one
    int next_round;
    const double* Cin;             // matrix before `round`
    double* Cout;                  // matrix after it
    double* V;                     // rotations of `round` (read by both halves)
    const double* D;               // rotated diagonal tiles of `round`
    double* Vnext;
    double* Dnext;
    double* off_next;              // pivot weight accumulator of the sweep the next round belongs to
};
__global__ void __launch_bounds__(512) la_fused_kernel(int ld, int nb, LaRound q, double* __restrict__ X, size_t mat_stride) {
    __shared__ double sm[6 * BT * LS];
    __shared__ double2 rot[2][BH];
    const int np = nb / 2;
    if ((int)blockIdx.x < np)
        la_solve_body<false>(sm, rot, blockIdx.x, ld, nb, q.next_round, q.round, q.full, q.Cin, q.V, q.D, q.Vnext, q.Dnext, q.off_next,
                             mat_stride);
    else
        block_jacobi_round_body<5>(sm, rot, (int)blockIdx.x - np, ld, nb, q.round, 0, q.Cin, q.Cout, X, q.off_next, mat_stride, q.V, q.D);
}

// The captured sweeps of one batch size.  A stream alternates between sizes (a whole signal: groups of eight hops, a ragged last
// group, then single hops again), and a capture is ~100 launches at n = 800: the workspace keeps the sets of the last few sizes.
struct GevdLargeGraphs {
    int batch = 0;
    int last_sweeps = 0;     // sweeps the last converged call of this size took
    // [0] two sweeps C0 -> C0 (the stretch nobody tests), [1] one sweep C0 -> C1, [2] one sweep C1 -> C0
    hipGraph_t graph[3] = {nullptr, nullptr, nullptr};
    hipGraphExec_t exec[3] = {nullptr, nullptr, nullptr};
    void destroy() {
        for (int g = 0; g < 3; ++g) {
            if (exec[g]) (void)hipGraphExecDestroy(exec[g]);
            if (graph[g]) (void)hipGraphDestroy(graph[g]);
            exec[g] = nullptr;
            graph[g] = nullptr;
        }
    }
};

struct GevdLargeWs {
    int n = 0, cap = 0;      // order, and the batch the buffers were sized for (smaller batches use their head)
    double *Bw = nullptr, *W = nullptr, *T1 = nullptr, *C0 = nullptr, *C1 = nullptr, *X = nullptr, *Li = nullptr;
    double *acc = nullptr, *coef = nullptr, *Vbuf = nullptr, *Vbuf2 = nullptr, *Dbuf = nullptr, *Dbuf2 = nullptr;

    int *flag = nullptr, *order = nullptr;
    std::vector<GevdLargeGraphs> sets;       // most recently used last; at most kMaxSets
    static constexpr size_t kMaxSets = 4;
    GevdLargeGraphs& set_for(int batch) {
        for (size_t i = 0; i < sets.size(); ++i)
            if (sets[i].batch == batch) {
                std::rotate(sets.begin() + i, sets.begin() + i + 1, sets.end());
                return sets.back();
            }
        if (sets.size() == kMaxSets) {
            sets.front().destroy();
            sets.erase(sets.begin());
        }
        sets.emplace_back();
        sets.back().batch = batch;
        return sets.back();
    }
    void release() {
        for (GevdLargeGraphs& g : sets) g.destroy();
        void* bufs[] = {Bw, W, T1, C0, C1, X, Li, acc, coef, Vbuf, Vbuf2, Dbuf, Dbuf2, flag, order};

        for (void* b : bufs)
            if (b) (void)hipFree(b);
        *this = GevdLargeWs();
    }
};

void apv_gevd_large_free(apv_handle* h) {
    if (h->gl_ws) {
        static_cast<GevdLargeWs*>(h->gl_ws)->release();
        delete static_cast<GevdLargeWs*>(h->gl_ws);
        h->gl_ws = nullptr;
    }
}

// Everything above, for `batch` independent pairs.  d_A, d_B: [batch][n][n] f64 (row-major, device, B is loaded
// with +reg on its diagonal here, reg x d_reg_scale[z] when that device array is given); outputs d_U [batch][n][n] (sorted columns), d_lam [batch][n]; optional
// d_r [batch][n] -> d_w [batch][V][n] for the ranks d_ranks[0..V) (device; nullptr = 1..V).  h_status[batch]: 0 ok, 1 not positive definite, 2 sweep cap.
int apv_gevd_large(apv_handle* h, int n, int batch, const double* d_A, const double* d_B, double reg,
                   const double* d_reg_scale, double* d_U, double* d_lam, const double* d_r, double mu, int V, const int* d_ranks, double* d_w, int32_t* h_status) {
    hipStream_t st = h->stream;
    const auto t_begin = std::chrono::steady_clock::now();
    // the caller consumes the leading lead_rank eigenpairs only (apvast.py:406-414): kernels_gevd_lead.hip, with this function's
    // block Jacobi as the fall-back.  One-shot request, like gl_tol2.
    const int lead_rank = h->gl_lead_rank;
    h->gl_lead_rank = 0;
    h->gl_lead_done = 0;
    const int ne = (n + BT - 1) / BT * BT, ld = ne;          // padded with ghost rows/columns: zero in A and C, unit in B
    const int nbk = ne / BT, nb = ne / BH, np = nb / 2, rounds = nb - 1;
    const size_t ms = (size_t)ne * ne, vs = (size_t)ne;
#define LCHK(call)                                                                                   \
    do {                                                                                             \
        hipError_t _e = (call);                                                                      \
        if (_e != hipSuccess) return apv_fail(h, APV_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)
    if (!h->gl_ws) h->gl_ws = new GevdLargeWs();
    GevdLargeWs& ws = *static_cast<GevdLargeWs*>(h->gl_ws);
    const size_t mb = sizeof(double) * ms * batch;
    const int gx = (ne + TPB - 1) / TPB;
    const int tiles = np * (np + 1) / 2;
    // the pair solves and the updates of a round as two launches (see the kernel); APV_LARGE_SPLIT=0 is the A/B switch back to one.
    // Measured (tools/bench_broadband.py): n = 256 two pairs 4.43 -> 4.29 ms per hop, sixteen pairs 1.24 -> 0.95 ms per hop,
    // n = 800 two pairs 32.6 -> 24.9 ms per hop -- the split wins even where the chip is far from full.
    static const int split_env = getenv("APV_LARGE_SPLIT") ? atoi(getenv("APV_LARGE_SPLIT")) : -1;
    const bool split = split_env != 0;
    if (ws.n != n || ws.cap < batch) {
        ws.release();
        ws.n = n;
        ws.cap = batch;
        LCHK(hipMalloc((void**)&ws.Bw, mb)); LCHK(hipMalloc((void**)&ws.W, mb)); LCHK(hipMalloc((void**)&ws.T1, mb));
        LCHK(hipMalloc((void**)&ws.C0, mb)); LCHK(hipMalloc((void**)&ws.C1, mb)); LCHK(hipMalloc((void**)&ws.X, mb));
        LCHK(hipMalloc((void**)&ws.Li, sizeof(double) * BT * BT * nbk * batch));
        LCHK(hipMalloc((void**)&ws.acc, sizeof(double) * 3 * batch));         // [sweep a | sweep b | ||C||_F^2]
        LCHK(hipMalloc((void**)&ws.coef, sizeof(double) * vs * batch));
        LCHK(hipMalloc((void**)&ws.Vbuf, sizeof(double) * (size_t)BT * LS * np * batch));
        LCHK(hipMalloc((void**)&ws.Vbuf2, sizeof(double) * (size_t)BT * LS * np * batch));
        LCHK(hipMalloc((void**)&ws.Dbuf, sizeof(double) * (size_t)BT * LS * np * batch));
        LCHK(hipMalloc((void**)&ws.Dbuf2, sizeof(double) * (size_t)BT * LS * np * batch));

        LCHK(hipMalloc((void**)&ws.flag, sizeof(int) * batch));
        LCHK(hipMalloc((void**)&ws.order, sizeof(int) * vs * batch));
    }
    GevdLargeGraphs& gs = ws.set_for(batch);
    // two sweeps: 2 (nb - 1) block rounds bring the ping-pong buffers back to where they started
    static const bool memset_node = getenv("APV_GRAPH_MEMSET") != nullptr;      // A/B switch: a memset node instead (see zero_f64_kernel)
    // look-ahead (see la_solve_kernel): APV_LARGE_LOOKAHEAD=0 is the A/B switch back to solve-then-update on one stream
    static const bool la_off = getenv("APV_LARGE_LOOKAHEAD") && atoi(getenv("APV_LARGE_LOOKAHEAD")) == 0;
    const bool lookahead = split && !la_off && np >= 2;
    // nsweeps = 2: both accumulators, C0 -> C0.  nsweeps = 1: one sweep from C0 (odd = false, accumulator a) or from C1 (odd = true,
    // accumulator b) into the other buffer -- rounds = nb - 1 is odd, so a sweep leaves the matrix in the buffer it did not start in.
    auto sweeps_la = [&](int nsweeps, bool odd) {
        double* const acc0 = ws.acc + (odd ? (size_t)batch : 0);
        hipLaunchKernelGGL(zero_f64_kernel, dim3((nsweeps * batch + 63) / 64), dim3(64), 0, st, nsweeps * batch, acc0);
        double *Cc = odd ? ws.C1 : ws.C0, *Cn = odd ? ws.C0 : ws.C1;
        const int total = nsweeps * rounds;
        // the pair solves of the very first round read the matrix as it stands
        hipLaunchKernelGGL(la_solve_kernel<true>, dim3(np, 1, batch), dim3(512), 0, st, ld, nb, 0, 0, 1, Cc, (const double*)ws.Vbuf2,
                           (const double*)ws.Dbuf2, ws.Vbuf, ws.Dbuf, acc0, ms);
        for (int gr = 0; gr < total; ++gr) {
            const int r = gr % rounds;
            double* Vcur = (gr & 1) ? ws.Vbuf2 : ws.Vbuf;
            double* Dcur = (gr & 1) ? ws.Dbuf2 : ws.Dbuf;
            double* Vnext = (gr & 1) ? ws.Vbuf : ws.Vbuf2;
            double* Dnext = (gr & 1) ? ws.Dbuf : ws.Dbuf2;
            if (gr + 1 < total) {
                const int rn = (gr + 1) % rounds, swn = (gr + 1) / rounds;
                LaRound q{r, rn == 0 ? 1 : 0, rn, Cc, Cn, Vcur, Dcur, Vnext, Dnext, acc0 + (size_t)swn * batch};
                hipLaunchKernelGGL(la_fused_kernel, dim3(np + tiles, 1, batch), dim3(512), 0, st, ld, nb, q, ws.X, ms);
            } else {
                hipLaunchKernelGGL(block_jacobi_round_kernel<5>, dim3(tiles, 1, batch), dim3(512), 0, st, ld, nb, r, 0, Cc, Cn, ws.X,
                                   acc0, ms, Vcur, (const double*)Dcur);
            }
            double* t = Cc; Cc = Cn; Cn = t;
        }
    };
    auto sweeps = [&](int nsweeps, bool odd) {
        if (lookahead) return sweeps_la(nsweeps, odd);
        double* const acc0 = ws.acc + (odd ? (size_t)batch : 0);
        if (memset_node) (void)hipMemsetAsync(acc0, 0, sizeof(double) * nsweeps * batch, st);
        else hipLaunchKernelGGL(zero_f64_kernel, dim3((nsweeps * batch + 63) / 64), dim3(64), 0, st, nsweeps * batch, acc0);
        double *Cc = odd ? ws.C1 : ws.C0, *Cn = odd ? ws.C0 : ws.C1;
        for (int sw = 0; sw < nsweeps; ++sw)
            for (int r = 0; r < rounds; ++r) {
                if (!split) {
                    hipLaunchKernelGGL(block_jacobi_round_kernel<0>, dim3(tiles, 1, batch), dim3(512), 0, st, ld, nb, r, r == 0 ? 1 : 0,
                                       Cc, Cn, ws.X, acc0 + (size_t)sw * batch, ms, ws.Vbuf, (const double*)nullptr);
                } else {
                    hipLaunchKernelGGL(block_jacobi_round_kernel<1>, dim3(np, 1, batch), dim3(512), 0, st, ld, nb, r, r == 0 ? 1 : 0,
                                       Cc, Cn, ws.X, acc0 + (size_t)sw * batch, ms, ws.Vbuf, (const double*)nullptr);
                    if (np > 1)
                        hipLaunchKernelGGL(block_jacobi_round_kernel<2>, dim3(tiles - np, 1, batch), dim3(512), 0, st, ld, nb, r, 0,
                                           Cc, Cn, ws.X, acc0 + (size_t)sw * batch, ms, ws.Vbuf, (const double*)nullptr);
                }
                double* t = Cc; Cc = Cn; Cn = t;
            }
    };
    static const bool no_graph = getenv("APV_NO_GRAPH") != nullptr;      // plain launches: rocprofv3 can then trace the rounds
    if (!gs.exec[0] && !no_graph) {
        for (int g = 0; g < 3; ++g) {
            LCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            sweeps(g == 0 ? 2 : 1, g == 2);
            // the capture is always closed, whatever was recorded: a stream left in capture mode fails every later call
            const hipError_t ce = hipStreamEndCapture(st, &gs.graph[g]);
            const hipError_t le = hipGetLastError();
            if (ce != hipSuccess || le != hipSuccess || !gs.graph[g]) {
                gs.destroy();
                return apv_fail(h, APV_ERR_HIP, std::string("capturing the Jacobi sweeps: ") + hipGetErrorString(ce != hipSuccess ? ce : le));
            }
            const hipError_t ie = hipGraphInstantiate(&gs.exec[g], gs.graph[g], nullptr, nullptr, 0);
            if (ie != hipSuccess) {
                // a set with exec[0] alone would skip the capture next time and launch null graphs: all or nothing
                gs.destroy();
                return apv_fail(h, APV_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie));
            }
        }
    }
    LCHK(hipMemsetAsync(ws.W, 0, mb, st));
    LCHK(hipMemsetAsync(ws.X, 0, mb, st));
    LCHK(hipMemsetAsync(ws.flag, 0, sizeof(int) * batch, st));
    LCHK(hipMemsetAsync(ws.acc, 0, sizeof(double) * 3 * batch, st));
    hipLaunchKernelGGL(load_pair_kernel, dim3(gx, ne, batch), dim3(TPB), 0, st, n, ne, ld, d_A, d_B, reg, d_reg_scale, ws.C0, ws.Bw, ms);   // C0 holds A for now
    static const bool old_panel = getenv("APV_LARGE_OLDPANEL") != nullptr;       // A/B switch: the 32-step elimination with a barrier a step
    static const bool panel_dbg = getenv("APV_LARGE_DEBUG") && atoi(getenv("APV_LARGE_DEBUG")) >= 2;
    if (panel_dbg) {
        const int on = 1;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_panel_stamps_on), &on, sizeof(int));
    }
    for (int k = 0; k < nbk; ++k)
        hipLaunchKernelGGL((old_panel ? chol_panel_kernel<false> : chol_panel_kernel<true>), dim3(nbk - k, 1, batch), dim3(256), 0, st, ld, k,
                           nbk, ws.Bw, ws.Li, ws.flag, ms);
    // APV_LARGE_OLDPRE=1: round 3's tile walk for W = L^-1 and its 32 x 32-tile products (A/B switch)
    if (panel_dbg) {
        (void)hipStreamSynchronize(st);
        unsigned long long hs[64 * 4];
        (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_panel_stamps), sizeof(hs));
        for (int k = 0; k < nbk - 1 && k < 64; k += (nbk > 12 ? 4 : 1))
            fprintf(stderr, "[apv gevd_large] panel %d, last row tile, s_memtime ticks: left-looking terms %llu, diagonal block %llu, product + store %llu\n", k,
                    hs[4 * k + 1] - hs[4 * k], hs[4 * k + 2] - hs[4 * k + 1], hs[4 * k + 3] - hs[4 * k + 2]);
    }
    static const bool old_pre = getenv("APV_LARGE_OLDPRE") != nullptr;
    if (old_pre) {
        hipLaunchKernelGGL(tri_inverse_kernel, dim3(nbk, BT / 8, batch), dim3(256), 0, st, ld, nbk, ws.Bw, ws.Li, ws.W, ws.flag, ms);
        const dim3 gg((n + BT - 1) / BT, (n + BT - 1) / BT, batch);
        hipLaunchKernelGGL((gemm_kernel<false, false>), gg, dim3(256), 0, st, n, ld, ws.W, ws.C0, ws.T1, ms);     // T1 = W A
        hipLaunchKernelGGL((gemm_kernel<false, true>), gg, dim3(256), 0, st, n, ld, ws.T1, ws.W, ws.C0, ms);      // C = T1 W^T
        hipLaunchKernelGGL(symmetrise_kernel, dim3(gx, n, batch), dim3(TPB), 0, st, n, ld, ws.C0, ms);
    } else {
        // W = L^-1 by recursive halving: with L = [L11 0; L21 L22], L^-1 = [W11 0; -W22 L21 W11  W22].  The diagonal 32 x 32
        // blocks are the panel kernel's; every level doubles the block (32 -> 64 -> ...): two products per level over all pairs of
        // the level at once (T1 is the scratch for L21 W11), ~ 2 log2(n / 32) launches instead of a walk of n / 32 dependent
        // block rows.  Ghost rows: the padded part of L is the identity, so is W's.
        hipLaunchKernelGGL(diag_inverse_scatter_kernel, dim3(nbk, 1, batch), dim3(256), 0, st, ld, nbk, ws.Li, ws.W, ms, ws.flag);
        for (int sz = BT; sz < ne; sz *= 2) {
            const int span = 2 * sz, full = ne / span, rem = ne - full * span;        // `full` complete pairs, then maybe a ragged one
            for (int part = 0; part < 2; ++part) {
                const int np_ = part == 0 ? full : (rem > sz ? 1 : 0);
                if (np_ == 0) continue;
                const int o0 = part == 0 ? 0 : full * span;                              // first index of the part's first pair
                const int m2 = part == 0 ? sz : rem - sz;                                 // rows of the second block
                const size_t off11 = (size_t)o0 * ld + o0, off21 = (size_t)(o0 + sz) * ld + o0, off22 = (size_t)(o0 + sz) * ld + o0 + sz;
                const size_t ps = (size_t)span * ld + span;
                Gemm64 g1{m2, sz, sz, ws.Bw + off21, ld, ms, ps, ws.W + off11, ld, ms, ps, ws.T1 + off21, ld, ms, ps, 1.0, np_, 0, 0, ws.flag};
                hipLaunchKernelGGL((gemm64_kernel<false>), dim3((sz + 63) / 64, (m2 + 63) / 64, batch * np_), dim3(256), 0, st, g1);   // T = L21 W11
                Gemm64 g2{m2, sz, m2, ws.W + off22, ld, ms, ps, ws.T1 + off21, ld, ms, ps, ws.W + off21, ld, ms, ps, -1.0, np_, 1, 0, ws.flag};
                hipLaunchKernelGGL((gemm64_kernel<false>), dim3((sz + 63) / 64, (m2 + 63) / 64, batch * np_), dim3(256), 0, st, g2);   // W21 = -W22 T
            }
        }
        const int gt = (ne + 63) / 64;
        Gemm64 ga{ne, ne, ne, ws.W, ld, ms, 0, ws.C0, ld, ms, 0, ws.T1, ld, ms, 0, 1.0, 1, 1, 0, ws.flag};
        hipLaunchKernelGGL((gemm64_kernel<false>), dim3(gt, gt, batch), dim3(256), 0, st, ga);                  // T1 = W A (W lower triangular)
        Gemm64 gb{ne, ne, ne, ws.T1, ld, ms, 0, ws.W, ld, ms, 0, ws.C0, ld, ms, 0, 1.0, 1, 2, 1, ws.flag};
        hipLaunchKernelGGL((gemm64_kernel<true>), dim3(gt, gt, batch), dim3(256), 0, st, gb);                   // C = T1 W^T, tiles on and below the diagonal
        hipLaunchKernelGGL(mirror_lower_kernel, dim3(gx, ne, batch), dim3(TPB), 0, st, ne, ld, ws.C0, ms, ws.flag);
    }
    hipLaunchKernelGGL(transpose_kernel, dim3(gx, n, batch), dim3(TPB), 0, st, n, ld, ws.W, ws.X, ms);         // X = W^T Q, Q = I
    std::vector<int> hflag(batch, 0);
    std::vector<double> hacc(3 * batch, 0.0);
    static const bool timing = getenv("APV_BB_TIMING") != nullptr;       // profiling aid, see stream_bb.hip
    // (a caller who sets a sweep cap or a sweep tolerance of his own is asking for the Jacobi iteration they belong to)
    int lead_b = (lead_rank > 0 && h->gl_tol2 <= 0.0 && h->cfg.max_sweeps <= 0) ? apv_gevd_lead_block(n, lead_rank) : 0, lead_done = 0;
    bool flags_read = false;
    if (timing) (void)hipStreamSynchronize(st);        // the timer's stage boundary (the product path does not stop here)
    const auto t_pre = std::chrono::steady_clock::now();
    if (lead_b > 0) {
        // C0 (whitened, symmetric) and X = W^T are read only; on *done == 0 nothing was written and the sweeps below run.  The
        // factorisation's flags come back with the first pass's results (no synchronisation of their own on this path).
        LCHK(hipMemcpyAsync(hflag.data(), ws.flag, sizeof(int) * batch, hipMemcpyDeviceToHost, st));
        const int lrc = apv_gevd_lead(h, n, ne, batch, lead_b, lead_rank, ws.C0, ws.X, d_U, d_lam, hflag.data(), &lead_done);
        if (lrc != APV_OK && lrc != APV_ERR_NOT_PD) return lrc;
        flags_read = true;
    }
    if (!lead_done) {
        hipLaunchKernelGGL(frob2_kernel, dim3(64, 1, batch), dim3(TPB), 0, st, n, ld, ws.C0, ws.acc + 2 * batch, ms);
        if (!flags_read) LCHK(hipMemcpyAsync(hflag.data(), ws.flag, sizeof(int) * batch, hipMemcpyDeviceToHost, st));
        LCHK(hipMemcpyAsync(hacc.data(), ws.acc, sizeof(double) * 3 * batch, hipMemcpyDeviceToHost, st));
        LCHK(hipStreamSynchronize(st));
    }
    bool any_bad = false;
    for (int z = 0; z < batch; ++z) {
        h_status[z] = hflag[z] ? 1 : 0;
        any_bad = any_bad || hflag[z];
    }
    if (any_bad) {
        // a failed factorisation leaves W undefined: nothing downstream may consume it
        (void)hipGetLastError();
        return apv_fail(h, APV_ERR_NOT_PD, "Matrix is not positive definite");
    }
    if (lead_done) {
        h->gl_lead_done = 1;
        if (d_r != nullptr && d_w != nullptr && V > 0) {
            hipLaunchKernelGGL(coef_kernel, dim3((lead_b + 31) / 32, 1, batch), dim3(TPB), 0, st, n, d_U, d_lam, d_r, mu, ws.coef, (size_t)n * n, vs);
            hipLaunchKernelGGL(vast_prefix_kernel, dim3(gx, 1, batch), dim3(TPB), 0, st, n, V, d_ranks, d_U, ws.coef, d_w, (size_t)n * n, vs,
                               (size_t)V * n);
        }
        // no synchronisation here: whoever consumes d_U / d_lam / d_w does so on this stream (the hop's output stage, a copy back)
        if (timing) LCHK(hipStreamSynchronize(st));
        LCHK(hipGetLastError());
        if (timing)
            fprintf(stderr, "[apv gevd_large] n=%d batch=%d: factor+whiten %.3f ms, leading %d of block %d %.3f ms\n", n, batch,
                    std::chrono::duration<double, std::milli>(t_pre - t_begin).count(), lead_rank, lead_b,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_pre).count());
        return APV_OK;
    }
    int n_sweeps = 0;
    std::vector<double> norm2(hacc.begin() + 2 * batch, hacc.end());
    const int max_sweeps = ((h->cfg.max_sweeps > 0 ? h->cfg.max_sweeps : 30) / 2 + 1) * 2;
    // a sweep whose pivots weigh <= tol ||C||^2 leaves ~tol^2 behind (quadratic convergence); APV_LARGE_TOL2 is a tuning aid
    static const double kLargeTol2 = getenv("APV_LARGE_TOL2") ? atof(getenv("APV_LARGE_TOL2")) : 1e-16;   // 1e-20 (round 1) cost two more sweeps for the same G1 errors
    // APV_LARGE_PAIRS=1: the stop test after every second sweep only, as before round 3's last change (A/B switch)
    static const bool pairs_only = getenv("APV_LARGE_PAIRS") && atoi(getenv("APV_LARGE_PAIRS")) != 0;
    bool converged = false;
    size_t last_off = 0;          // where in hacc the last tested sweep's pivot weights are
    // The stop test costs a copy and a host synchronisation.  Consecutive calls of a stream solve problems of the same kind: of
    // the sweeps the LAST call of this shape needed, all but the last three (an even count: they go as captured pairs) are
    // launched back to back before the first test; from there every sweep is tested, so that the sweep found to be the last one
    // IS the last one (tested in pairs, half the calls ran a sixteenth sweep after a fifteenth that had already met the bound).
    const int untested = (gs.last_sweeps > 4 && h->gl_tol2 <= 0.0 && !timing) ? ((gs.last_sweeps - 3) & ~1) : 0;
    const double tol2 = h->gl_tol2 > 0.0 ? h->gl_tol2 : kLargeTol2;
    while (n_sweeps < max_sweeps && !converged) {
        const bool odd = n_sweeps & 1;
        const bool pair = !odd && (n_sweeps < untested || pairs_only);
        if (gs.exec[0]) LCHK(hipGraphLaunch(gs.exec[pair ? 0 : (odd ? 2 : 1)], st));
        else sweeps(pair ? 2 : 1, odd);
        n_sweeps += pair ? 2 : 1;
        if (n_sweeps <= untested && n_sweeps < max_sweeps) continue;
        LCHK(hipMemcpyAsync(hacc.data(), ws.acc, sizeof(double) * 2 * batch, hipMemcpyDeviceToHost, st));
        LCHK(hipStreamSynchronize(st));
        last_off = (pair || odd) ? (size_t)batch : 0;
        const double* const wt = hacc.data() + last_off;      // the pivot weights of the sweep just run
        if (timing)
            for (int z = 0; z < batch; ++z)
                fprintf(stderr, "[apv gevd_large] sweep %d matrix %d: pivot weight / ||C||^2 %.2e\n", n_sweeps, z, wt[z] / norm2[z]);
        converged = true;
        for (int z = 0; z < batch; ++z)
            if (!(wt[z] <= tol2 * norm2[z])) converged = false;
    }
    double* const Cfin = (n_sweeps & 1) ? ws.C1 : ws.C0;
    const auto t_sweeps = std::chrono::steady_clock::now();
    if (converged && h->gl_tol2 <= 0.0) gs.last_sweeps = n_sweeps;
    if (!converged) {
        // only the members that are still above the bound carry the sweep-cap status (a batch sweeps until its slowest member is done)
        const double* const wl = hacc.data() + last_off;
        for (int z = 0; z < batch; ++z)
            if (!(wl[z] <= tol2 * norm2[z])) h_status[z] = 2;
    }
    hipLaunchKernelGGL(rank_kernel, dim3(gx, 1, batch), dim3(TPB), 0, st, n, ld, Cfin, d_lam, ws.order, ms, vs);
    hipLaunchKernelGGL(gather_cols_kernel, dim3(gx, n, batch), dim3(TPB), 0, st, n, ld, ws.X, ws.order, d_U, ms, vs,
                       (size_t)n * n);
    if (d_r != nullptr && d_w != nullptr && V > 0) {
        hipLaunchKernelGGL(coef_kernel, dim3((n + 31) / 32, 1, batch), dim3(TPB), 0, st, n, d_U, d_lam, d_r, mu, ws.coef, (size_t)n * n, vs);
        hipLaunchKernelGGL(vast_prefix_kernel, dim3(gx, 1, batch), dim3(TPB), 0, st, n, V, d_ranks, d_U, ws.coef, d_w, (size_t)n * n, vs,
                           (size_t)V * n);
    }
    LCHK(hipStreamSynchronize(st));
    LCHK(hipGetLastError());
    if (timing)
        fprintf(stderr, "[apv gevd_large] n=%d batch=%d: factor+whiten %.3f ms, %d sweeps %.3f ms, sort+filter %.3f ms\n", n, batch,
                std::chrono::duration<double, std::milli>(t_pre - t_begin).count(), n_sweeps,
                std::chrono::duration<double, std::milli>(t_sweeps - t_pre).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_sweeps).count());
#undef LCHK
    return APV_OK;
}
