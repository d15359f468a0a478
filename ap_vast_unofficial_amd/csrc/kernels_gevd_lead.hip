// Leading eigenpairs of the whitened matrix of a broadband pair: what the per-hop path consumes.
//
//   apvast.py:406-414 builds the filters from the first V generalised eigenpairs only (V = 8 of n = 256 at BASELINE config 1,
//   50 of 800 with the parameters of make_python_test.m:6-15); the full lambda_* / U_* (apvast.py:380-387) are attributes read
//   after the fact.  apv_gevd_large's block Jacobi diagonalises all of C = W A W^T: 12-15 sweeps x (n/16 - 1) dependent launches.
//   This file finds the leading b = V + guard eigenpairs of the same C by Chebyshev-filtered subspace iteration:
//
//     X0      deterministic pseudo-random block, n x b.  No state is carried from hop to hop: in the NumPy model
//             (tools/probes/lead_model.py, profiles/r04/lead_model.txt) a start from the previous hop's block saves a pass in
//             three of fourteen pairs at cfg1 (75 % of the statistics window shared) and nothing at the reference's
//             parameters (20 % shared), and a stateless solve keeps a resumed stream bit for bit
//     pass    Z = C Y                                         one block product (f64 MFMA, the matrix streams from L2 / HBM)
//             G = Y^T Y, H = Y^T Z                            b x b Gram matrices, K split over slabs of 128 rows
//             (H, G) -> T, theta                              ONE workgroup: scale to a unit diagonal, eliminate [G | I] (Cholesky
//                                                             factor and its inverse in one go), M = L^-1 H L^-T, cyclic Jacobi
//                                                             in LDS, sort, T = D L^-T Q
//             X = Y T, res_j = ||Z T_j - theta_j X_j||        Ritz vectors, their residuals (partial sums per row tile; the host
//                                                             adds them up) and step 1 of the next filter (C X = Z T is known)
//             host: converged?  else degree m of the next filter from the Ritz values
//             Y = p_m(C) X                                    scaled Chebyshev polynomial that damps [0, theta_b] (C is positive
//                                                             semi-definite): m - 1 products
//     end     U[:, :b] = W^T X, lambda[:b] = theta            jdiag's contract on the leading columns (apvast.py:31-35)
//
//   The degree is capped by the amplification ratio T_m(x(theta_1)) between the largest Ritz value and the damped interval:
//   1e6 in the first filter, 1e10 in the second, 1e12 from there -- beyond that the guard columns turn into copies of the
//   leading eigenvector and the Gram matrix loses rank (model: cond 1e16 at degree 16 with lambda_1 / lambda_V = 8).
//   A pair whose leading V residuals do not reach the bound within kMaxPass passes (a flat spectrum: lambda_{b+1} / lambda_V
//   close to one), or whose Gram matrix breaks down, hands the whole call back to the block Jacobi of apv_gevd_large.
#include "apv_internal.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {

using d4 = __attribute__((ext_vector_type(4))) double;

constexpr int LEAD_MAXB = 32;            // matrices per call
constexpr int kMaxPass = 12;             // Rayleigh-Ritz passes before the call falls back

struct LeadCoef {                        // Ynew = a C Ycur + b Ycur + g Yprev, per matrix
    double a[LEAD_MAXB], b[LEAD_MAXB], g[LEAD_MAXB];
    unsigned active;                     // bit z: matrix z takes part in this launch
    const double* dev;                   // non-null: the coefficients are dev[(2 z + dev_step) * 3 + {0, 1, 2}] (lead_bounds_kernel's)
    int dev_step;
};

__device__ __forceinline__ double lead_rnd(unsigned i) {
    unsigned x = i * 0x9E3779B9u + 0x7F4A7C15u;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return (double)(int)x * (1.0 / 2147483648.0);
}

// X[z][i][j] = pseudo-random in (-1, 1) for i < n, 0 on the ghost rows of the padding
__global__ void __launch_bounds__(256) lead_init_kernel(int n, int ne, int b, double* __restrict__ X, size_t y_stride) {
    X += blockIdx.z * y_stride;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= ne * b) return;
    X[idx] = (idx / b) < n ? lead_rnd((unsigned)idx) : 0.0;
}

// Yo = a (A Yc) + b Yc + g Yp for the block Yc [ne][b] (row-major).  A workgroup owns 16 rows x 16 NB columns; the K dimension
// is dealt to its four waves in chunks of 16 (one 32-byte load per lane of A's rows: lane (r, kq) holds A[r][k0 + 4 kq .. + 3],
// and MFMA j of the chunk contracts the indices k0 + 4 kq + j -- any order of the contraction index serves, as long as the B
// operand follows it), the four partial tiles meet in LDS in a fixed order.
template <int NB, int NWV = 4>
__global__ void __launch_bounds__(64 * NWV) lead_mult_kernel(int ne, int lda, size_t a_stride, const double* __restrict__ A, int b,
                                                        size_t y_stride, const double* __restrict__ Yc,
                                                        const double* __restrict__ Yp, double* __restrict__ Yo, int rows_out,
                                                        int ldo, size_t o_stride, LeadCoef cf) {
    __shared__ double red[NWV][NB][256];
    const int z = blockIdx.z;
    if (!((cf.active >> z) & 1u)) return;
    const double ca = cf.dev ? cf.dev[(2 * z + cf.dev_step) * 3] : cf.a[z], cb = cf.dev ? cf.dev[(2 * z + cf.dev_step) * 3 + 1] : cf.b[z],
                 cg = cf.dev ? cf.dev[(2 * z + cf.dev_step) * 3 + 2] : cf.g[z];
    A += z * a_stride;
    Yc += z * y_stride;
    Yo += z * o_stride;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, kq = lane >> 4;
    // Workgroups go to the eight XCDs round robin by their linear index, and each XCD has an L2 of its own: the column tiles of one
    // row tile get indices that are congruent modulo 8, so that A's 16-row strip is fetched into ONE L2, by the first of them --
    // and, the mapping being the same in every launch, is still there for the next product (two matrices of order 800 are 1.4 MB
    // per XCD).  blockIdx.x runs over row tiles x column tiles.
    const int nct = gridDim.y == 1 ? (int)(b / (16 * NB)) : 1;
    int rt, ct;
    {
        const int id = blockIdx.x, nrt8 = (int)(gridDim.x / nct) / 8 * 8, grp = id / (8 * nct);
        if (grp * 8 < nrt8) { rt = grp * 8 + (id & 7); ct = (id >> 3) % nct; }
        else { const int rem = id - nrt8 * nct; rt = nrt8 + rem / nct; ct = rem % nct; }       // the ragged last row tiles
    }
    const int row0 = rt * 16, col0 = ct * 16 * NB;
    d4 acc[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) acc[c] = d4{0, 0, 0, 0};
    if (ca != 0.0) {
        const double* arow = A + (size_t)(row0 + il) * lda + 4 * kq;
        const double* ycol = Yc + (size_t)(4 * kq) * b + col0 + il;
        // two chunks of 16 in flight per wave: the loads of the next pair go out before the MFMAs of this one
        constexpr int PF = 2;
        d4 av[PF];
        double bv[PF][4][NB];
        auto fetch = [&](int k0, int u) {
            if (k0 < ne) {
                av[u] = *reinterpret_cast<const d4*>(arow + k0);
                const double* yr = ycol + (size_t)k0 * b;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < NB; ++c) bv[u][j][c] = yr[(size_t)j * b + 16 * c];
            }
        };
#pragma unroll
        for (int u = 0; u < PF; ++u) fetch(w * 16 + 16 * NWV * u, u);
        for (int k0 = w * 16; k0 < ne; k0 += 16 * NWV * PF) {
            d4 cav[PF];
            double cbv[PF][4][NB];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                cav[u] = av[u];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < NB; ++c) cbv[u][j][c] = bv[u][j][c];
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) fetch(k0 + 16 * NWV * (PF + u), u);
#pragma unroll
            for (int u = 0; u < PF; ++u)
                if (k0 + 16 * NWV * u < ne) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int c = 0; c < NB; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(cav[u][j], cbv[u][j][c], acc[c], 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int c = 0; c < NB; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) red[w][c][t * 64 + lane] = acc[c][t];
    __syncthreads();
    // wave w < 4 finishes accumulator row t = w: element (row0 + kq + 4 w, col0 + 16 c + il); the partials are added in wave order
    if (w >= 4) return;
    const int row = row0 + kq + 4 * w;
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        const int col = col0 + 16 * c + il;
        double s = ((red[0][c][w * 64 + lane] + red[1][c][w * 64 + lane]) + red[2][c][w * 64 + lane]) + red[3][c][w * 64 + lane];
#pragma unroll
        for (int x = 4; x < NWV; ++x) s += red[x][c][w * 64 + lane];
        double v = ca * s;
        if (cb != 0.0) v = __builtin_fma(cb, Yc[(size_t)row * b + col], v);
        if (cg != 0.0) v = __builtin_fma(cg, Yp[z * y_stride + (size_t)row * b + col], v);
        if (row < rows_out) Yo[(size_t)row * ldo + col] = v;
    }
}

// Bounds for the FIRST filter, before any Ritz value exists: c = trace(C) / n -- on the oracle's pairs 0.29-0.61 of lambda_{b+1}
// (tools/probes/lead_model.py): below the block, so nothing wanted is damped -- and s1 = max_j sum_i |C_ij| >= lambda_1 (3-6 x it).
// The filter [0, c] of degree 1 or 2 (amplification T_m(2 s1 / c - 1) <= 1e6) takes the place of a Rayleigh-Ritz pass on the
// random start block, which gave no better bounds (its smallest Ritz value is about the mean too) and cost a projected
// eigenproblem and a host synchronisation.  coef[z][2][3]: the two steps' (a, b, g); fixed-order sums: the same bits every run.
// part[z][chunk][j] = sum over the chunk's rows i of |C_ij| (column j = row j); 8 row chunks
constexpr int LEAD_BCH = 8;
__global__ void __launch_bounds__(256) lead_colsum_kernel(int n, int ld, const double* __restrict__ C, size_t mat_stride,
                                                          double* __restrict__ part) {
    const int z = blockIdx.z, ch = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    C += z * mat_stride;
    const int rows = (n + LEAD_BCH - 1) / LEAD_BCH, i0 = ch * rows, i1 = min(n, i0 + rows);
    double cs = 0.0;
#pragma unroll 8
    for (int i = i0; i < i1; ++i) cs += fabs(C[(size_t)i * ld + j]);
    part[((size_t)z * LEAD_BCH + ch) * n + j] = cs;
}

__global__ void __launch_bounds__(1024) lead_bounds_kernel(int n, int ld, const double* __restrict__ C, size_t mat_stride,
                                                           const double* __restrict__ part, double* __restrict__ coef) {
    __shared__ double red[1024], redm[1024];
    const int z = blockIdx.x, tid = threadIdx.x;
    C += z * mat_stride;
    part += (size_t)z * LEAD_BCH * n;
    double tr = 0.0, mx = 0.0;
    for (int j = tid; j < n; j += 1024) {
        tr += C[(size_t)j * ld + j];
        double cs = 0.0;
#pragma unroll
        for (int ch = 0; ch < LEAD_BCH; ++ch) cs += part[(size_t)ch * n + j];
        mx = cs > mx ? cs : mx;
    }
    red[tid] = tr;
    redm[tid] = mx;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if (tid < w) {
            red[tid] += red[tid + w];
            redm[tid] = redm[tid] > redm[tid + w] ? redm[tid] : redm[tid + w];
        }
        __syncthreads();
    }
    if (tid == 0) {
        double s1 = redm[0], c = red[0] / (double)n;
        if (!(s1 > 0.0)) s1 = 1.0;
        if (!(c > 1e-12 * s1)) c = 1e-12 * s1;
        if (c > 0.5 * s1) c = 0.5 * s1;
        const double x1 = 2.0 * s1 / c - 1.0;
        const int m = acosh(1e6) / acosh(x1) >= 2.0 ? 2 : 1;
        const double e = 0.5 * c, sigma1 = e / (s1 - e);
        double* o = coef + (size_t)z * 6;
        o[0] = sigma1 / e; o[1] = -sigma1; o[2] = 0.0;                                  // Y1 = (sigma_1 / e) (C - e) X
        if (m == 2) {
            const double sn = 1.0 / (2.0 / sigma1 - sigma1);
            o[3] = 2.0 * sn / e; o[4] = -2.0 * sn; o[5] = -sigma1 * sn;                 // Y2 = (2 sigma_2 / e) (C - e) Y1 - sigma_1 sigma_2 X
        } else {
            o[3] = 0.0; o[4] = 1.0; o[5] = 0.0;                                         // degree 1: the second step copies
        }
    }
}

// Partial Gram matrices of one slab of 128 rows: tile (ti, tj), tj <= ti, of G = Y^T Y and H = Y^T Z.  Gp / Hp:
// [z][slab][pair][256], element (4 t + (lane >> 4), lane & 15) of the tile at t * 64 + lane.
__global__ void __launch_bounds__(256) lead_gram_kernel(int ne, int b, size_t y_stride, const double* __restrict__ Y,
                                                        const double* __restrict__ Z, double* __restrict__ Gp,
                                                        double* __restrict__ Hp, unsigned active) {
    __shared__ double red[2][4][256];
    const int z = blockIdx.z, s = blockIdx.y;
    if (!((active >> z) & 1u)) return;
    int ti = 0, tj = blockIdx.x;
    while (tj > ti) { tj -= ti + 1; ++ti; }
    Y += z * y_stride;
    Z += z * y_stride;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, kq = lane >> 4;
    d4 ag = {0, 0, 0, 0}, ah = {0, 0, 0, 0};
    const int r0 = s * 128 + w * 32;
    if (r0 < ne) {
#pragma unroll
        for (int k0 = 0; k0 < 32; k0 += 4) {
            const size_t ro = (size_t)(r0 + k0 + kq) * b;
            const double a = Y[ro + 16 * ti + il], bg = Y[ro + 16 * tj + il], bh = Z[ro + 16 * tj + il];
            ag = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bg, ag, 0, 0, 0);
            ah = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bh, ah, 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        red[0][w][t * 64 + lane] = ag[t];
        red[1][w][t * 64 + lane] = ah[t];
    }
    __syncthreads();
    const size_t o = (((size_t)z * gridDim.y + s) * gridDim.x + blockIdx.x) * 256 + tid;
    Gp[o] = ((red[0][0][tid] + red[0][1][tid]) + red[0][2][tid]) + red[0][3][tid];
    Hp[o] = ((red[1][0][tid] + red[1][1][tid]) + red[1][2][tid]) + red[1][3][tid];
}

// 1/sqrt(x) to full double precision: v_rsq_f64 and one third-order correction (as in kernels_gevd_large.hip)
__device__ __forceinline__ double lead_rsq(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
}

// Jacobi rotation J = [[c, s], [-s, c]] that annihilates beta in [[alpha, beta], [beta, gamma]]; tangent in float, (c, s) formed
// in double from it (orthogonal to 1e-16, leaves ~1e-7 |beta| behind: the sweeps stay quadratic).  See sym_rotation in
// kernels_gevd_large.hip.
__device__ __forceinline__ void lead_rotation(double alpha, double gamma, double beta, double& c, double& s) {
    const double b2 = beta * beta;
    const bool rotate = b2 > 1e-290 && b2 > 1e-60 * (alpha * alpha + gamma * gamma);
    const double d = gamma - alpha, tb = 2.0 * beta;
    const int ex = __builtin_amdgcn_frexp_exp(fabs(d) >= fabs(tb) ? d : tb);
    const float fd = (float)__builtin_ldexp(d, -ex), fb = (float)__builtin_ldexp(tb, -ex);
    const float s2 = __builtin_fmaf(fd, fd, fb * fb);
    const float hyp = s2 * __builtin_amdgcn_rsqf(s2);
    const float mag = fabsf(fb) * __builtin_amdgcn_rcpf(fabsf(fd) + hyp);
    const float t = __builtin_copysignf(mag, __builtin_copysignf(1.0f, fd) * fb);
    const double td = (double)t;
    const double cc = lead_rsq(__builtin_fma(td, td, 1.0));
    c = rotate ? cc : 1.0;
    s = rotate ? td * cc : 0.0;
}

// 16 x 16 tile of op(A) op(B) with operands in LDS (row stride ld), K deep: acc[t] = element (row0 + (lane >> 4) + 4 t, col0 + (lane & 15))
template <bool TA, bool TB, int K>
__device__ __forceinline__ d4 lead_tile(const double* A, const double* Bm, int ld, int row0, int col0, int lane) {
    const int il = lane & 15, kq = lane >> 4;
    d4 acc = {0, 0, 0, 0};
#pragma unroll 4
    for (int k0 = 0; k0 < K; k0 += 4) {
        const double av = TA ? A[(k0 + kq) * ld + row0 + il] : A[(row0 + il) * ld + k0 + kq];
        const double bv = TB ? Bm[(col0 + il) * ld + k0 + kq] : Bm[(k0 + kq) * ld + col0 + il];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
    return acc;
}

// ONE wave: D (order 16, symmetric, in LDS with row stride ld) -> W = L^-1 with D = L L^T, written to Wout (lower triangle, zeros
// above).  [D | I] is eliminated with the rows in registers: lane = row i + 16 x column group cq holds D[i][4 cq ..] and the right
// half's [i][4 cq ..]; a step hands the pivot column of the left half (= its pivot row: what is left is symmetric) and the pivot
// row of the right half round through 32 doubles of LDS -- no barrier, the LDS executes a wave's accesses in order (the scheme
// of kernels_gevd64.hip stage 1 and of chol_panel_kernel).  Returns true when a pivot is <= floor or not finite.
__device__ __forceinline__ bool lead_wave_inv16(const double* D, double* Wout, int ld, double* buf, int lane, double floor) {
    const int i = lane & 15, cq = lane >> 4;
    double b[4], w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        b[u] = D[i * ld + 4 * cq + u];
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
        bad = bad || !(dq > floor) || !(dq < 1e300);
        double inv = __builtin_amdgcn_rcp(dq);
        inv = inv * __builtin_fma(-dq, inv, 2.0);
        inv = inv * __builtin_fma(-dq, inv, 2.0);
        if (i == q) dsc = lead_rsq(dq);
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
    for (int u = 0; u < 4; ++u) Wout[i * ld + 4 * cq + u] = (4 * cq + u <= i) ? w[u] * dsc : 0.0;
    return bad;
}

// one wave, acc += fa(., k) fb(k, .) over 16 values of k on v_mfma_f64_16x16x4_f64: acc[t] = element (kq + 4 t, il)
template <typename FA, typename FB>
__device__ __forceinline__ d4 lead_mm16(FA fa, FB fb, int lane, d4 acc) {
    const int il = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa(il, k0 + kq), fb(k0 + kq, il), acc, 0, 0, 0);
    return acc;
}

// The projected problem of one pass: (H, G) of order B -> T [B][B] (X = Y T has orthonormal columns that diagonalise C on
// span Y), theta [B] descending; hinfo[z] = {theta[B], Gram breakdown, sweeps, Jacobi met its bound, -} is what the host reads.
template <int B, int NT, bool MOVE>
__global__ void __launch_bounds__(NT) lead_small_kernel(int nslab, const double* __restrict__ Gp, const double* __restrict__ Hp,
                                                        int max_sweeps, double tol2, double* __restrict__ T,
                                                        double* __restrict__ theta, double* __restrict__ hinfo, unsigned active,
                                                        unsigned long long* __restrict__ stamps) {
    constexpr int LD = B + 1, NTL = B / 16, NPAIR = NTL * (NTL + 1) / 2, NP = B / 2, NW = NT / 64;
    extern __shared__ double sm[];
    double* S0 = sm;                 // G -> W H -> Q
    double* S1 = S0 + B * LD;        // H -> M
    double* S2 = S1 + B * LD;        // W = L^-1
    double* S3 = S2 + B * LD;        // MOVE: the second buffer of M's ping-pong (S2 is Q's once W^T has moved into S0)
    double* dsc = S3 + (MOVE ? B * LD : 0);       // [B] 1/sqrt(diag G)
    double* th = dsc + B;            // [B]
    double* pw = th + B;             // [NP] pivot weights of a sweep
    double2* cs = reinterpret_cast<double2*>(pw + NP + (NP & 1));    // [NP]
    int* pq = reinterpret_cast<int*>(cs + NP);                       // [B]: p of pair k, then q of pair k
    int* rk = pq + B;                                                // [B]
    __shared__ int fail;
    __shared__ double scal[2];
    const int z = blockIdx.x;
    if (!((active >> z) & 1u)) return;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    // diagnostics (APV_LEAD_DEBUG=2): s_memtime of thread 0 at the phase boundaries, eight per matrix
    auto stamp = [&](int i) { if (stamps != nullptr && tid == 0) stamps[8 * z + i] = __builtin_amdgcn_s_memtime(); };
    stamp(0);
    Gp += (size_t)z * nslab * NPAIR * 256;
    Hp += (size_t)z * nslab * NPAIR * 256;
    if (tid == 0) fail = 0;
    // a: the slabs' partial sums, in slab order; strictly lower tiles are mirrored
    for (int e = tid; e < NPAIR * 256; e += NT) {
        const int pair = e >> 8, rem = e & 255, t = rem >> 6, ln = rem & 63;
        int ti = 0, tj = pair;
        while (tj > ti) { tj -= ti + 1; ++ti; }
        // (four slabs' loads go out together, the additions stay in slab order: one trip to memory per four slabs instead of one per slab)
        double g = 0.0, hh = 0.0;
        for (int s0 = 0; s0 < nslab; s0 += 4) {
            double gv[4], hv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool in = s0 + u < nslab;
                gv[u] = in ? Gp[((size_t)(s0 + u) * NPAIR + pair) * 256 + rem] : 0.0;
                hv[u] = in ? Hp[((size_t)(s0 + u) * NPAIR + pair) * 256 + rem] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (s0 + u < nslab) {
                    g += gv[u];
                    hh += hv[u];
                }
        }
        const int i = 16 * ti + (ln >> 4) + 4 * t, j = 16 * tj + (ln & 15);
        S0[i * LD + j] = g;
        S1[i * LD + j] = hh;
        if (ti != tj) {
            S0[j * LD + i] = g;
            S1[j * LD + i] = hh;
        }
    }
    __syncthreads();
    // diagonal tiles of H: Y^T Z is symmetric up to rounding only
    for (int e = tid; e < NTL * 120; e += NT) {
        const int tl = e / 120;
        int i = 1, j = e % 120;
        while (j >= i) { j -= i; ++i; }          // 0 <= j < i < 16
        const int gi = 16 * tl + i, gj = 16 * tl + j;
        const double m = 0.5 * (S1[gi * LD + gj] + S1[gj * LD + gi]);
        S1[gi * LD + gj] = m;
        S1[gj * LD + gi] = m;
    }
    if (tid < B) {
        const double g = S0[tid * LD + tid];
        if (!(g > 0.0)) fail = 1;
        dsc[tid] = g > 0.0 ? lead_rsq(g) : 1.0;
    }
    __syncthreads();
    // b: unit diagonal (the columns of Y differ by the filter's amplification, up to 1e12), W = I
    for (int e = tid; e < B * B; e += NT) {
        const int i = e / B, j = e % B;
        const double sc = dsc[i] * dsc[j];
        S0[i * LD + j] *= sc;
        S1[i * LD + j] *= sc;
        S2[i * LD + j] = i == j ? 1.0 : 0.0;
    }
    __syncthreads();
    stamp(1);
    if constexpr (MOVE) {
        // c: W = L^-1 with G = L L^T, blocked in 16 x 16 tiles as the order-64 kernel's factorisation: the diagonal tile is inverted by
        // ONE wave (lead_wave_inv16, 16 steps without a barrier), the panel L_ik = G_ik W_kk^T and the trailing update
        // G_ij -= L_ik L_jk^T are tile products on the matrix cores, and the off-diagonal tiles of W follow by block forward
        // substitution, W_ba = -W_bb sum_k L_bk W_ka, one block diagonal after the other: 5 B / 16 - 2 barriers instead of B, and
        // no pass over all of both halves per pivot (the loop below: 28 us of the kernel's 103 at B = 64).  The pivots are those of
        // the unblocked elimination; one at or below 1e-13 (the Gram matrix has a unit diagonal) is a breakdown.
        const int il = lane & 15, kq = lane >> 4;
        for (int kb = 0; kb < NTL; ++kb) {
            const int o = 16 * kb;
            if (wv == 0) {
                const bool bad = lead_wave_inv16(S0 + o * LD + o, S2 + o * LD + o, LD, th, lane, 1e-13);
                if (bad && lane == 0) fail = 1;
            }
            __syncthreads();
            if (fail || kb == NTL - 1) break;                          // (uniform: a shared word, read after the barrier)
            for (int ib = kb + 1 + wv; ib < NTL; ib += NW) {
                const double* const Gik = S0 + (16 * ib) * LD + o;
                const double* const Wkk = S2 + o * LD + o;
                const d4 t = lead_mm16([&](int i, int kk) { return Gik[i * LD + kk]; }, [&](int kk, int j) { return Wkk[j * LD + kk]; }, lane,
                                       d4{0.0, 0.0, 0.0, 0.0});
#pragma unroll
                for (int u = 0; u < 4; ++u) S0[(16 * ib + kq + 4 * u) * LD + o + il] = t[u];      // only this wave reads this tile
            }
            __syncthreads();
            const int rem = NTL - 1 - kb, ntile = rem * (rem + 1) / 2;
            for (int tl = wv; tl < ntile; tl += NW) {
                int ib = kb + 1, w2 = tl;
                while (w2 > ib - kb - 1) {                                 // tiles (kb+1,kb+1), (kb+2,kb+1), (kb+2,kb+2), ...
                    w2 -= ib - kb;
                    ++ib;
                }
                const int jb = kb + 1 + w2;
                const double* const Lik = S0 + (16 * ib) * LD + o;
                const double* const Ljk = S0 + (16 * jb) * LD + o;
                const d4 t = lead_mm16([&](int i, int kk) { return Lik[i * LD + kk]; }, [&](int kk, int j) { return Ljk[j * LD + kk]; }, lane,
                                       d4{0.0, 0.0, 0.0, 0.0});
#pragma unroll
                for (int u = 0; u < 4; ++u) S0[(16 * ib + kq + 4 * u) * LD + 16 * jb + il] -= t[u];
            }
            __syncthreads();
        }
        if (fail) {
            if (tid == 0) {
                hinfo[(size_t)z * (B + 4) + B] = 1.0;
                hinfo[(size_t)z * (B + 4) + B + 1] = 0.0;
                hinfo[(size_t)z * (B + 4) + B + 2] = 0.0;
            }
            return;
        }
        for (int dg = 1; dg < NTL; ++dg) {
            for (int a = wv; a + dg < NTL; a += NW) {
                const int bb = a + dg;
                d4 acc = {0.0, 0.0, 0.0, 0.0};
                for (int k = a; k < bb; ++k) {
                    const double* const Lbk = S0 + (16 * bb) * LD + 16 * k;
                    const double* const Wka = S2 + (16 * k) * LD + 16 * a;
                    acc = lead_mm16([&](int i, int kk) { return Lbk[i * LD + kk]; }, [&](int kk, int j) { return Wka[kk * LD + j]; }, lane, acc);
                }
                double* const Wba = S2 + (16 * bb) * LD + 16 * a;        // the sum first, as the second product's operand
#pragma unroll
                for (int u = 0; u < 4; ++u) Wba[(kq + 4 * u) * LD + il] = acc[u];
                const double* const Wbb = S2 + (16 * bb) * LD + 16 * bb;
                const d4 t = lead_mm16([&](int i, int kk) { return Wbb[i * LD + kk]; }, [&](int kk, int j) { return Wba[kk * LD + j]; }, lane,
                                       d4{0.0, 0.0, 0.0, 0.0});
#pragma unroll
                for (int u = 0; u < 4; ++u) Wba[(kq + 4 * u) * LD + il] = -t[u];
            }
            __syncthreads();
        }
    } else {
    // c: eliminate [G | I]: G = Lt D Lt^T, the right half becomes Lt^-1; pivots stay on G's diagonal
    // (a register-resident variant -- rows in registers, pivot row and column published through LDS, the 64 steps unrolled --
    // was measured: 8 256 instructions, and SLOWER, 178 949 ticks against 66 928 at B = 64; this loop stays)
    for (int k = 0; k < B - 1; ++k) {
        const double piv = S0[k * LD + k];
        if (!(piv > 1e-13)) {                    // uniform: every thread reads the same word
            if (tid == 0) fail = 1;
            break;
        }
        const double rp = 1.0 / piv;
        for (int e = tid; e < (B - 1 - k) * B; e += NT) {
            const int i = k + 1 + e / B, j = e % B;
            const double f = S0[i * LD + k] * rp;
            if (j > k) S0[i * LD + j] = __builtin_fma(-f, S0[k * LD + j], S0[i * LD + j]);
            else S2[i * LD + j] = __builtin_fma(-f, S2[k * LD + j], S2[i * LD + j]);
        }
        __syncthreads();
    }
    __syncthreads();
    if (!(S0[(B - 1) * LD + B - 1] > 1e-13) && tid == 0) fail = 1;
    __syncthreads();
    if (fail) {
        if (tid == 0) {
            hinfo[(size_t)z * (B + 4) + B] = 1.0;
            hinfo[(size_t)z * (B + 4) + B + 1] = 0.0;
            hinfo[(size_t)z * (B + 4) + B + 2] = 0.0;
        }
        return;
    }
    if (tid < B) th[tid] = lead_rsq(S0[tid * LD + tid]);
    __syncthreads();
    for (int e = tid; e < B * B; e += NT) S2[(e / B) * LD + e % B] *= th[e / B];        // W = D^-1/2 Lt^-1
    __syncthreads();
    }
    stamp(2);
    // d: M = W H W^T
    for (int tile = wv; tile < NTL * NTL; tile += NW) {
        const int r0 = (tile / NTL) * 16, c0 = (tile % NTL) * 16;
        const d4 a = lead_tile<false, false, B>(S2, S1, LD, r0, c0, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) S0[(r0 + (lane >> 4) + 4 * t) * LD + c0 + (lane & 15)] = a[t];
    }
    __syncthreads();
    for (int tile = wv; tile < NTL * NTL; tile += NW) {
        const int r0 = (tile / NTL) * 16, c0 = (tile % NTL) * 16;
        const d4 a = lead_tile<false, true, B>(S0, S2, LD, r0, c0, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) S1[(r0 + (lane >> 4) + 4 * t) * LD + c0 + (lane & 15)] = a[t];
    }
    __syncthreads();
    double nrm = 0.0;
    for (int e = tid; e < B * B; e += NT) {
        const int i = e / B, j = e % B;
        if (j < i) {
            const double m = 0.5 * (S1[i * LD + j] + S1[j * LD + i]);
            S1[i * LD + j] = m;
            S1[j * LD + i] = m;
            nrm += 2.0 * m * m;
        } else if (j == i) {
            const double m = S1[i * LD + i];
            nrm += m * m;
        }
        // Q = I (W H is spent); MOVE: the accumulator starts as W^T, so that the sweeps leave W^T Q and W's buffer is free for them
        S0[i * LD + j] = MOVE ? S2[j * LD + i] : (i == j ? 1.0 : 0.0);
    }
    // ||M||_F^2 (order of the sum fixed: lanes through DPP-free LDS tree)
    __shared__ double redn[NT];
    redn[tid] = nrm;
    __syncthreads();
    for (int len = NT; len > 1;) {                     // (NT = 768 is not a power of two)
        const int half = (len + 1) >> 1;
        if (tid < len - half) redn[tid] += redn[tid + half];
        len = half;
        __syncthreads();
    }
    const double norm2 = redn[0];
    stamp(3);
    // e: cyclic Jacobi, round-robin pairing (player 0 fixed, the others rotate)
    int sweeps = 0, conv = 0;
    double *Mc = S1, *Qc = S0;                 // where M and the accumulator are when the sweeps are done
    if constexpr (MOVE) {
        // The same tournament with the ROWS AND COLUMNS MOVING instead of the pairing: pair k is always slots (k, B-1-k), and a
        // round writes what it rotated one slot on (slot 0 stays, s -> s + 1, B-1 -> 1: the circle method), M and the accumulator
        // each into a second buffer.  Which elements a thread reads and where it writes them is then the same in every round --
        // no pairing table, no index arithmetic inside the round, the addresses are loop invariants -- and the rotations are
        // those of the in-place form bit for bit (same pairs, same order of operands).  After B - 1 rounds everything is back in
        // its own slot.  The round is bound by the instructions the one compute unit issues: the in-place form spent ~150 per
        // thread and round at B = 64 (pair indices and rotations fetched from LDS, two multiply-adds per address).
        auto nx = [](int sl) { return sl == 0 ? 0 : (sl == B - 1 ? 1 : sl + 1); };
        double *Mn = S3, *Qn = S2;
        constexpr int MI = (NP * NP + NT - 1) / NT, QI = (B * NP + NT - 1) / NT;
        // loop-invariant element offsets of this thread's blocks
        int m_rd[MI][4], m_wr[MI][4], m_k[MI], m_l[MI], q_rd[QI][2], q_wr[QI][2], q_l[QI];
#pragma unroll
        for (int it = 0; it < MI; ++it) {
            const int e = tid + it * NT, k = e / NP, l = e - k * NP;
            const int pk = k, qk = B - 1 - k, pl = l, ql = B - 1 - l;
            m_k[it] = e < NP * NP ? k : -1;
            m_l[it] = l;
            m_rd[it][0] = pk * LD + pl; m_rd[it][1] = pk * LD + ql; m_rd[it][2] = qk * LD + pl; m_rd[it][3] = qk * LD + ql;
            m_wr[it][0] = nx(pk) * LD + nx(pl); m_wr[it][1] = nx(pk) * LD + nx(ql);
            m_wr[it][2] = nx(qk) * LD + nx(pl); m_wr[it][3] = nx(qk) * LD + nx(ql);
        }
#pragma unroll
        for (int it = 0; it < QI; ++it) {
            const int e = tid + it * NT, i = e / NP, l = e - i * NP;
            q_l[it] = e < B * NP ? l : -1;
            q_rd[it][0] = i * LD + l; q_rd[it][1] = i * LD + B - 1 - l;
            q_wr[it][0] = i * LD + nx(l); q_wr[it][1] = i * LD + nx(B - 1 - l);
        }
        for (int sw = 0; sw < max_sweeps && !conv; ++sw) {
            double pwk = 0.0;
            for (int r = 0; r < B - 1; ++r) {
                if (tid < NP) {
                    const int p = tid, q = B - 1 - tid;
                    const double al = Mc[p * LD + p], ga = Mc[q * LD + q], be = Mc[p * LD + q];
                    double c, sn;
                    lead_rotation(al, ga, be, c, sn);
                    pwk = __builtin_fma(be, be, pwk);
                    cs[tid] = double2{c, sn};
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < MI; ++it) {
                    if (m_k[it] < 0) continue;
                    const double2 rk2 = cs[m_k[it]], rl = cs[m_l[it]];
                    const double a = Mc[m_rd[it][0]], b_ = Mc[m_rd[it][1]], c_ = Mc[m_rd[it][2]], d_ = Mc[m_rd[it][3]];
                    const double a1 = a * rl.x - b_ * rl.y, b1 = a * rl.y + b_ * rl.x;
                    const double c1 = c_ * rl.x - d_ * rl.y, d1 = c_ * rl.y + d_ * rl.x;
                    Mn[m_wr[it][0]] = rk2.x * a1 - rk2.y * c1;
                    Mn[m_wr[it][1]] = rk2.x * b1 - rk2.y * d1;
                    Mn[m_wr[it][2]] = rk2.y * a1 + rk2.x * c1;
                    Mn[m_wr[it][3]] = rk2.y * b1 + rk2.x * d1;
                }
#pragma unroll
                for (int it = 0; it < QI; ++it) {
                    if (q_l[it] < 0) continue;
                    const double2 rl = cs[q_l[it]];
                    const double x = Qc[q_rd[it][0]], y = Qc[q_rd[it][1]];
                    Qn[q_wr[it][0]] = x * rl.x - y * rl.y;
                    Qn[q_wr[it][1]] = x * rl.y + y * rl.x;
                }
                __syncthreads();
                double* t = Mc; Mc = Mn; Mn = t;
                t = Qc; Qc = Qn; Qn = t;
            }
            if (tid < NP) pw[tid] = pwk;
            __syncthreads();
            if (tid == 0) {
                double t = 0.0;
                for (int k = 0; k < NP; ++k) t += pw[k];
                scal[0] = t;
            }
            __syncthreads();
            ++sweeps;
            conv = scal[0] <= tol2 * norm2;
            __syncthreads();
        }
    } else {
    for (int sw = 0; sw < max_sweeps && !conv; ++sw) {
        double pwk = 0.0;
        // (Tried: every thread forms the two rotations its 2 x 2 block needs from the pivots it reads itself -- one barrier and no
        // rotation table per round.  Slower: 1.00 -> 1.63 us a round at B = 64.  The round is bound by the instructions the one
        // compute unit issues, not by its three LDS round trips, and sixteen waves forming rotations redundantly are ~60 more
        // vector instructions per thread and round.  The table stays.)
        for (int r = 0; r < B - 1; ++r) {
            if (tid < NP) {
                const int s1 = tid, s2 = B - 1 - tid;
                int a1 = s1 - 1 - r; a1 %= (B - 1); if (a1 < 0) a1 += B - 1;
                int a2 = s2 - 1 - r; a2 %= (B - 1); if (a2 < 0) a2 += B - 1;
                const int p = s1 == 0 ? 0 : 1 + a1, q = 1 + a2;
                const double al = S1[p * LD + p], ga = S1[q * LD + q], be = S1[p * LD + q];
                double c, s;
                lead_rotation(al, ga, be, c, s);
                pwk = __builtin_fma(be, be, pwk);
                cs[tid] = double2{c, s};
                pq[tid] = p;
                pq[NP + tid] = q;
            }
            __syncthreads();
            for (int e = tid; e < NP * NP; e += NT) {
                const int k = e / NP, l = e % NP;
                const int pk = pq[k], qk = pq[NP + k], pl = pq[l], ql = pq[NP + l];
                const double2 rk2 = cs[k], rl = cs[l];
                const double a = S1[pk * LD + pl], b_ = S1[pk * LD + ql], c_ = S1[qk * LD + pl], d_ = S1[qk * LD + ql];
                const double a1 = a * rl.x - b_ * rl.y, b1 = a * rl.y + b_ * rl.x;
                const double c1 = c_ * rl.x - d_ * rl.y, d1 = c_ * rl.y + d_ * rl.x;
                S1[pk * LD + pl] = rk2.x * a1 - rk2.y * c1;
                S1[pk * LD + ql] = rk2.x * b1 - rk2.y * d1;
                S1[qk * LD + pl] = rk2.y * a1 + rk2.x * c1;
                S1[qk * LD + ql] = rk2.y * b1 + rk2.x * d1;
            }
            for (int e = tid; e < B * NP; e += NT) {
                const int i = e / NP, l = e % NP;
                const int pl = pq[l], ql = pq[NP + l];
                const double2 rl = cs[l];
                const double x = S0[i * LD + pl], y = S0[i * LD + ql];
                S0[i * LD + pl] = x * rl.x - y * rl.y;
                S0[i * LD + ql] = x * rl.y + y * rl.x;
            }
            __syncthreads();
        }
        if (tid < NP) pw[tid] = pwk;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int k = 0; k < NP; ++k) t += pw[k];
            scal[0] = t;
        }
        __syncthreads();
        ++sweeps;
        conv = scal[0] <= tol2 * norm2;         // a sweep whose pivots weigh <= tol2 ||M||^2 leaves ~tol2^2 behind
        __syncthreads();
    }
    }
    // f: descending order, T = D W^T Q with its columns in that order
    stamp(4);
    if (tid < B) th[tid] = Mc[tid * LD + tid];
    __syncthreads();
    if (tid < B) {
        const double li = th[tid];
        int rank = 0;
        for (int j = 0; j < B; ++j) rank += (th[j] > li) || (th[j] == li && j < tid);
        rk[tid] = rank;
        theta[(size_t)z * B + rank] = li;
        hinfo[(size_t)z * (B + 4) + rank] = li;           // the host's copy: {theta[B], breakdown, sweeps, Jacobi met its bound, -}
    }
    __syncthreads();
    T += (size_t)z * B * B;
    if constexpr (MOVE) {
        // the accumulator started as W^T: it holds W^T Q
        for (int e = tid; e < B * B; e += NT) {
            const int i = e / B, j = e - i * B;
            T[(size_t)i * B + rk[j]] = dsc[i] * Qc[i * LD + j];
        }
    } else {
        for (int tile = wv; tile < NTL * NTL; tile += NW) {
            const int r0 = (tile / NTL) * 16, c0 = (tile % NTL) * 16;
            const d4 a = lead_tile<true, false, B>(S2, S0, LD, r0, c0, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int i = r0 + (lane >> 4) + 4 * t, j = c0 + (lane & 15);
                T[(size_t)i * B + rk[j]] = dsc[i] * a[t];
            }
        }
    }
    if (tid == 0) {
        hinfo[(size_t)z * (B + 4) + B] = 0.0;
        hinfo[(size_t)z * (B + 4) + B + 1] = (double)sweeps;
        hinfo[(size_t)z * (B + 4) + B + 2] = (double)conv;
    }
    stamp(5);
}

// X = Y T and, from CX = Z T, the residual partials respart[z][row tile][col] = sum over the tile's 16 rows of (CX - theta X)^2
// and the first step of the next filter, Y1 = (sigma_1 / e) (CX - e X) with the damped interval [0, c = theta_b] (e = c / 2,
// sigma_1 = e / (theta_1 - e)): C X is known here, so that step needs no product with C.
__global__ void __launch_bounds__(256) lead_rot_kernel(int ne, int b, size_t y_stride, const double* __restrict__ Y,
                                                       const double* __restrict__ Z, const double* __restrict__ T,
                                                       const double* __restrict__ theta, double* __restrict__ X,
                                                       double* __restrict__ Y1, double* __restrict__ respart, unsigned active) {
    __shared__ double red[2][4][256];
    __shared__ double r2[16][17];
    const int z = blockIdx.z;
    if (!((active >> z) & 1u)) return;
    Y += z * y_stride; Z += z * y_stride; X += z * y_stride; Y1 += z * y_stride;
    T += (size_t)z * b * b;
    theta += (size_t)z * b;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, il = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * 16, col0 = blockIdx.y * 16;
    d4 ax = {0, 0, 0, 0}, ac = {0, 0, 0, 0};
    const int kw = b / 4;
    for (int k0 = w * kw; k0 < (w + 1) * kw; k0 += 4) {
        const double ay = Y[(size_t)(row0 + il) * b + k0 + kq], az = Z[(size_t)(row0 + il) * b + k0 + kq];
        const double bt = T[(size_t)(k0 + kq) * b + col0 + il];
        ax = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, bt, ax, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f64_16x16x4f64(az, bt, ac, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        red[0][w][t * 64 + lane] = ax[t];
        red[1][w][t * 64 + lane] = ac[t];
    }
    __syncthreads();
    const int e = w * 64 + lane;
    const double x = ((red[0][0][e] + red[0][1][e]) + red[0][2][e]) + red[0][3][e];
    const double cx = ((red[1][0][e] + red[1][1][e]) + red[1][2][e]) + red[1][3][e];
    const int rl = kq + 4 * w, row = row0 + rl, col = col0 + il;
    X[(size_t)row * b + col] = x;
    const double th0 = theta[0];
    double c = theta[b - 1];
    if (!(c > 1e-12 * th0)) c = 1e-12 * th0;
    const double hc = 0.5 * c, a1 = 1.0 / (th0 - hc);            // sigma_1 / e
    Y1[(size_t)row * b + col] = a1 * (cx - hc * x);
    const double rr = __builtin_fma(-theta[col], x, cx);
    r2[rl][il] = rr * rr;
    __syncthreads();
    if (tid < 16) {
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += r2[r][tid];
        respart[((size_t)z * gridDim.x + blockIdx.x) * b + col0 + tid] = s;
    }
}

// lam[z][j] = theta[z][j], j < b (dense [batch][n] eigenvalue array of the caller)
__global__ void __launch_bounds__(64) lead_lam_kernel(int n, int b, const double* __restrict__ theta, double* __restrict__ lam) {
    const int z = blockIdx.x, j = threadIdx.x;
    if (j < b) lam[(size_t)z * n + j] = theta[(size_t)z * b + j];
}

struct LeadWs {
    int ne = 0, b = 0, cap = 0;
    double* P[3] = {nullptr, nullptr, nullptr};
    double *Zb = nullptr, *Gp = nullptr, *Hp = nullptr, *T = nullptr, *theta = nullptr;
    unsigned long long* stamps = nullptr;   // APV_LEAD_DEBUG=2: phase stamps of lead_small_kernel, [batch][8]
    double* coefdev = nullptr;        // [batch][2][3] the first filter's coefficients (lead_bounds_kernel)
    double* out = nullptr;            // what the host reads after a pass: [batch][row tiles][b] residual partials, then [batch][b + 4] hinfo
    double* h_out = nullptr;          // pinned
    void release() {
        void* bufs[] = {P[0], P[1], P[2], Zb, Gp, Hp, T, theta, out, coefdev, stamps};
        for (void* p : bufs)
            if (p) (void)hipFree(p);
        if (h_out) (void)hipHostFree(h_out);
        *this = LeadWs();
    }
};

template <int B, int NT, bool MOVE>
hipError_t launch_small(hipStream_t st, int batch, int nslab, const double* Gp, const double* Hp, int max_sweeps, double tol2,
                        double* T, double* theta, double* hinfo, unsigned active, unsigned long long* stamps) {
    constexpr int LD = B + 1, NP = B / 2;
    const size_t bytes = sizeof(double) * ((MOVE ? 4 : 3) * (size_t)B * LD + 2 * B + NP + (NP & 1)) + sizeof(double2) * NP + sizeof(int) * 2 * B;
    static bool once = false;
    if (!once) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lead_small_kernel<B, NT, MOVE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        once = true;
    }
    hipLaunchKernelGGL((lead_small_kernel<B, NT, MOVE>), dim3(batch), dim3(NT), bytes, st, nslab, Gp, Hp, max_sweeps, tol2, T, theta, hinfo,
                       active, stamps);
    return hipGetLastError();
}

}  // namespace

void apv_gevd_lead_free(apv_handle* h) {
    if (h->lead_ws) {
        static_cast<LeadWs*>(h->lead_ws)->release();
        delete static_cast<LeadWs*>(h->lead_ws);
        h->lead_ws = nullptr;
    }
}

// Block width for `rank` wanted eigenpairs of an order-n problem, 0 when the leading solver does not apply
int apv_gevd_lead_block(int n, int rank) {
    static const int off = getenv("APV_LEAD") ? atoi(getenv("APV_LEAD")) == 0 : 0;
    if (off || rank <= 0) return 0;
    int b = (rank + 16 + 15) / 16 * 16;
    if (b < 32) b = 32;
    if (b > 64) b = 64;
    if (b - rank < 8) return 0;              // too few guard vectors
    if (3 * b > n) return 0;                 // the block is no small part of the matrix: the full solve is the cheaper one
    return b;
}

// C: [batch][ne][ld = ne] whitened matrices (symmetric, zero ghost rows / columns), WT: [batch][ne][ne] = W^T.
// On success with *done = 1: d_U[z][i][j] (n x n, j < b) and d_lam[z][j] (j < b) hold the leading b eigenpairs.
// *done = 0: nothing was written, the caller runs the full solve.
// h_pd_flags[batch]: host words an asynchronous copy queued BEFORE this call fills with the factorisation's flags (non-zero: the
// dark matrix was not positive definite and C is garbage); they are valid after the first pass's synchronisation, and the call
// then returns APV_ERR_NOT_PD with *done = 0.
int apv_gevd_lead(apv_handle* h, int n, int ne, int batch, int b, int rank, const double* C, const double* WT, double* d_U,
                  double* d_lam, const int* h_pd_flags, int* done) {
    *done = 0;
    if (batch > LEAD_MAXB || b % 16 != 0 || b < 32 || b > 64 || ne % 32 != 0) return APV_OK;
    hipStream_t st = h->stream;
#define LCHK(call)                                                                                   \
    do {                                                                                             \
        hipError_t _e = (call);                                                                      \
        if (_e != hipSuccess) return apv_fail(h, APV_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)
    if (!h->lead_ws) h->lead_ws = new LeadWs();
    LeadWs& ws = *static_cast<LeadWs*>(h->lead_ws);
    const size_t ys = (size_t)ne * b, ms = (size_t)ne * ne;
    const int nrt = ne / 16, nslab = (ne + 127) / 128, ntl = b / 16, npair = ntl * (ntl + 1) / 2;
    const size_t n_part = (size_t)batch * nrt * b, ow = n_part + (size_t)batch * (b + 4);
    if (ws.ne != ne || ws.b != b || ws.cap < batch) {
        ws.release();
        ws.ne = ne; ws.b = b; ws.cap = batch;
        for (int i = 0; i < 3; ++i) LCHK(hipMalloc((void**)&ws.P[i], sizeof(double) * ys * batch));
        LCHK(hipMalloc((void**)&ws.Zb, sizeof(double) * ys * batch));
        LCHK(hipMalloc((void**)&ws.Gp, sizeof(double) * (size_t)batch * nslab * npair * 256));
        LCHK(hipMalloc((void**)&ws.Hp, sizeof(double) * (size_t)batch * nslab * npair * 256));
        LCHK(hipMalloc((void**)&ws.T, sizeof(double) * (size_t)batch * b * b));
        LCHK(hipMalloc((void**)&ws.theta, sizeof(double) * (size_t)batch * b));
        LCHK(hipMalloc((void**)&ws.coefdev, sizeof(double) * 6 * batch));
        LCHK(hipMalloc((void**)&ws.stamps, sizeof(unsigned long long) * 8 * batch));
        LCHK(hipMalloc((void**)&ws.out, sizeof(double) * ow));
        LCHK(hipHostMalloc((void**)&ws.h_out, sizeof(double) * ow));
    }
    static const bool dbg = getenv("APV_LEAD_DEBUG") != nullptr;
    static const bool dbg2 = dbg && atoi(getenv("APV_LEAD_DEBUG")) >= 2;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned active = batch >= 32 ? 0xFFFFFFFFu : ((1u << batch) - 1u);
    LeadCoef one{};
    one.active = active;
    for (int z = 0; z < batch; ++z) { one.a[z] = 1.0; one.b[z] = 0.0; one.g[z] = 0.0; }
    auto mult = [&](const double* A, const double* Yc, const double* Yp, double* Yo, int rows_out, int ldo, size_t os, const LeadCoef& cf) {
        // 16 x 16 outputs per workgroup while that fills the chip, 16 x 32 beyond
        const long wgs = (long)nrt * (b / 16) * batch;
        static const int kForceNB = getenv("APV_LEAD_NB") ? atoi(getenv("APV_LEAD_NB")) : 0;      // tuning aid
        // waves that split K in a 16 x 16 workgroup (tuning aid; measured at n = 800 / 256, whole solver: 4 -> 1.372 / 0.366 ms, 8 -> 1.339 /
        // 0.358, 16 -> 1.421 / 0.396: the product is not bound by what one wave has in flight)
        static const int kWaves = getenv("APV_LEAD_MULT_WAVES") ? atoi(getenv("APV_LEAD_MULT_WAVES")) : 8;
        if ((kForceNB == 2 || (kForceNB == 0 && wgs > 2048)) && b % 32 == 0)
            hipLaunchKernelGGL((lead_mult_kernel<2>), dim3(nrt * (b / 32), 1, batch), dim3(256), 0, st, ne, ne, ms, A, b, ys, Yc, Yp, Yo, rows_out, ldo, os, cf);
        else if (kWaves == 8)
            hipLaunchKernelGGL((lead_mult_kernel<1, 8>), dim3(nrt * (b / 16), 1, batch), dim3(512), 0, st, ne, ne, ms, A, b, ys, Yc, Yp, Yo, rows_out, ldo, os, cf);
        else if (kWaves == 16)
            hipLaunchKernelGGL((lead_mult_kernel<1, 16>), dim3(nrt * (b / 16), 1, batch), dim3(1024), 0, st, ne, ne, ms, A, b, ys, Yc, Yp, Yo, rows_out, ldo, os, cf);
        else
            hipLaunchKernelGGL((lead_mult_kernel<1>), dim3(nrt * (b / 16), 1, batch), dim3(256), 0, st, ne, ne, ms, A, b, ys, Yc, Yp, Yo, rows_out, ldo, os, cf);
    };
    int ic = 0, ix = 1, iy = 2;
    int final_buf[LEAD_MAXB];
    for (int z = 0; z < batch; ++z) final_buf[z] = -1;
    hipLaunchKernelGGL(lead_init_kernel, dim3((unsigned)((ys + 255) / 256), 1, batch), dim3(256), 0, st, n, ne, b, ws.P[ic], ys);
    // the first filter from trace and column-sum bounds (lead_bounds_kernel); APV_LEAD_PREFILTER=0: a Rayleigh-Ritz pass on the
    // random block instead, as until the middle of round 4 (A/B switch)
    static const bool prefilter = !(getenv("APV_LEAD_PREFILTER") && atoi(getenv("APV_LEAD_PREFILTER")) == 0);
    int total_mv = 0;
    if (prefilter) {
        hipLaunchKernelGGL(lead_colsum_kernel, dim3((n + 255) / 256, LEAD_BCH, batch), dim3(256), 0, st, n, ne, C, ms, ws.Zb);   // Zb is free here
        hipLaunchKernelGGL(lead_bounds_kernel, dim3(batch), dim3(1024), 0, st, n, ne, C, ms, (const double*)ws.Zb, ws.coefdev);
        LeadCoef d0 = one, d1 = one;
        d0.dev = d1.dev = ws.coefdev;
        d0.dev_step = 0;
        d1.dev_step = 1;
        mult(C, ws.P[0], nullptr, ws.P[1], ne, b, ys, d0);
        mult(C, ws.P[1], ws.P[0], ws.P[2], ne, b, ys, d1);
        total_mv += 2;
        ic = 2; ix = 0; iy = 1;
    }
    static const double kLimitsRR[3] = {1e6, 1e10, 1e12}, kLimitsPF[3] = {1e10, 1e12, 1e12};
    const double* const kLimits = prefilter ? kLimitsPF : kLimitsRR;
    // measured (tools/probes/lead_sweeps_probe.sh, profiles/r04/lead_sweeps_probe.txt): one sweep per non-final pass at b = 64
    // (n = 800: 1.61 ms for seven passes; two sweeps 1.77 for six, three 2.09, four 2.39), two at b = 32 (n = 256: 0.51 ms, the same
    // with one or three)
    static const int kSweepsEnv = getenv("APV_LEAD_SWEEPS") ? atoi(getenv("APV_LEAD_SWEEPS")) : 0;     // tuning aid
    const int kPartialSweeps = kSweepsEnv > 0 ? kSweepsEnv : (b >= 48 ? 1 : 2);
    int pass = 0;
    bool fallback = false;
    for (;; ++pass) {
        // Rayleigh-Ritz on span P[ic]
        one.active = active;
        mult(C, ws.P[ic], nullptr, ws.Zb, ne, b, ys, one);
        ++total_mv;
        hipLaunchKernelGGL(lead_gram_kernel, dim3(npair, nslab, batch), dim3(256), 0, st, ne, b, ys, ws.P[ic], ws.Zb, ws.Gp, ws.Hp, active);
        hipError_t se;
        // Any orthogonal Q serves: X = Y T has orthonormal columns whatever the sweeps leave undone, theta_j = M_jj is the Rayleigh
        // quotient of X_j, and the residual ||C X_j - theta_j X_j|| measured below certifies the pair by itself.  The sweeps only
        // have to keep the columns close to Ritz vectors (the filter's amplification then scales columns instead of making them
        // parallel) and sharpen the Ritz values that set the next filter's bounds; an off-diagonal element left in the leading
        // block shows in the residuals and costs another pass.
        static const int kSweepEvery = getenv("APV_LEAD_SWEEP_EVERY") ? atoi(getenv("APV_LEAD_SWEEP_EVERY")) : 1;      // tuning aid
        // APV_LEAD_SWEEP_SCHEDULE="a,b,c,...": sweeps of pass 0, 1, 2, ... (the last entry for all later passes); tuning aid
        static const std::vector<int> kSchedule = [] {
            std::vector<int> v;
            if (const char* e = getenv("APV_LEAD_SWEEP_SCHEDULE"))
                for (const char* q = e; *q;) {
                    v.push_back(atoi(q));
                    while (*q && *q != ',') ++q;
                    if (*q == ',') ++q;
                }
            return v;
        }();
        int msw = (kSweepEvery > 1 && pass > 1 && pass % kSweepEvery != 0) ? 0 : kPartialSweeps;
        if (!kSchedule.empty()) msw = kSchedule[pass < (int)kSchedule.size() ? pass : (int)kSchedule.size() - 1];
        double* const hinfo = ws.out + n_part;
        // sixteen waves for the one workgroup of a matrix and the sweeps in their moving-slot form (lead_small_kernel, e);
        // APV_LEAD_WIDE=0: 256 / 256 / 512 threads and the in-place sweeps with their pairing table, as in the middle of round 4
        static const bool kWide = getenv("APV_LEAD_WIDE") == nullptr || atoi(getenv("APV_LEAD_WIDE")) != 0;
        unsigned long long* const stp = dbg2 ? ws.stamps : nullptr;
        if (b == 32) se = kWide ? launch_small<32, 512, true>(st, batch, nslab, ws.Gp, ws.Hp, msw, 1e-18, ws.T, ws.theta, hinfo, active, stp)
                                : launch_small<32, 256, false>(st, batch, nslab, ws.Gp, ws.Hp, msw, 1e-18, ws.T, ws.theta, hinfo, active, stp);
        else if (b == 48) se = kWide ? launch_small<48, 768, true>(st, batch, nslab, ws.Gp, ws.Hp, msw, 1e-18, ws.T, ws.theta, hinfo, active, stp)
                                     : launch_small<48, 256, false>(st, batch, nslab, ws.Gp, ws.Hp, msw, 1e-18, ws.T, ws.theta, hinfo, active, stp);
        else se = kWide ? launch_small<64, 1024, true>(st, batch, nslab, ws.Gp, ws.Hp, msw, 1e-18, ws.T, ws.theta, hinfo, active, stp)
                        : launch_small<64, 512, false>(st, batch, nslab, ws.Gp, ws.Hp, msw, 1e-18, ws.T, ws.theta, hinfo, active, stp);
        LCHK(se);
        // Ritz vectors into P[ix], the next filter's first step into P[iy]; the residual partials are summed by the host (row tiles in order)
        hipLaunchKernelGGL(lead_rot_kernel, dim3(nrt, b / 16, batch), dim3(256), 0, st, ne, b, ys, ws.P[ic], ws.Zb, ws.T, ws.theta, ws.P[ix],
                           ws.P[iy], ws.out, active);
        LCHK(hipMemcpyAsync(ws.h_out, ws.out, sizeof(double) * ow, hipMemcpyDeviceToHost, st));
        // (polling the stream instead of the blocking wait was measured: no difference, the runtime's wait spins already)
        LCHK(hipStreamSynchronize(st));
        if (dbg2) {
            unsigned long long hs[8];
            (void)hipMemcpy(hs, ws.stamps, sizeof(hs), hipMemcpyDeviceToHost);
            fprintf(stderr, "[apv lead]   lead_small_kernel matrix 0, s_memtime ticks: load+scale %llu, eliminate %llu, whiten+norm %llu, sweeps %llu, sort+T %llu\n",
                    hs[1] - hs[0], hs[2] - hs[1], hs[3] - hs[2], hs[4] - hs[3], hs[5] - hs[4]);
        }
        if (pass == 0 && h_pd_flags)
            for (int z = 0; z < batch; ++z)
                if (h_pd_flags[z]) return APV_ERR_NOT_PD;
        LeadCoef step[16];
        int deg[LEAD_MAXB], mdeg = 0;
        for (int z = 0; z < batch; ++z) {
            deg[z] = 0;
            if (!((active >> z) & 1u)) continue;
            const double* th = ws.h_out + n_part + (size_t)z * (b + 4);
            const double* inf = th + b;
            if (inf[0] != 0.0 || !(th[0] > 0.0)) { fallback = true; break; }
            double rmax = 0.0;
            for (int j = 0; j < rank; ++j) {
                double s2 = 0.0;
                for (int t = 0; t < nrt; ++t) s2 += ws.h_out[((size_t)z * nrt + t) * b + j];
                const double rj = sqrt(s2);
                rmax = rj > rmax ? rj : rmax;
            }
            // The leading `rank` vectors span the invariant subspace to rmax / gap; the bound asks for 1e-10 there and 1e-12 of the
            // largest eigenvalue on every residual, but never for less than float64 gives (a cluster that straddles the cut has no
            // gap: any orthonormal basis of it is as good as the one LAPACK's rounding picks for the reference)
            const double gap = th[rank - 1] - th[rank];
            double target = 1e-12 * th[0];
            if (1e-10 * gap < target) target = 1e-10 * gap;
            const double floor_ = 3e-14 * sqrt((double)n) * th[0];
            if (target < floor_) target = floor_;
            const bool conv = rmax <= target;
            if (dbg)
                fprintf(stderr, "[apv lead] pass %d matrix %d: theta1 %.4e theta_V/theta1 %.3f theta_b/theta_V %.3f gap %.2e res %.2e target %.2e jacobi %d sweeps%s\n",
                        pass, z, th[0], th[rank - 1] / th[0], th[b - 1] / th[rank - 1], gap / th[0], rmax / th[0], target / th[0], (int)inf[1], conv ? " converged" : "");
            if (conv) {
                active &= ~(1u << z);
                final_buf[z] = ix;
                continue;
            }
            if (pass + 1 >= kMaxPass) { fallback = true; break; }
            // degree of the next filter
            double c = th[b - 1];
            if (!(c > 1e-12 * th[0])) c = 1e-12 * th[0];
            const double x1 = 2.0 * th[0] / c - 1.0;
            const double lim = kLimits[pass < 2 ? pass : 2];
            int m = (int)floor(acosh(lim) / acosh(x1 > 1.0 + 1e-12 ? x1 : 1.0 + 1e-12));
            const double xv = 2.0 * th[rank - 1] / c - 1.0;
            if (xv > 1.0 + 1e-9 && rmax > 0.0) {
                const int need = (int)ceil(log(10.0 * rmax / target) / acosh(xv));
                if (need >= 1 && need < m) m = need;        // as long as the bound asks, not as long as the cap allows
            }
            m = m < 1 ? 1 : (m > 16 ? 16 : m);
            if (dbg) fprintf(stderr, "[apv lead]   matrix %d: next filter of degree %d (x_1 = %.2f, x_V = %.3f)\n", z, m, x1, xv);
            deg[z] = m;
            mdeg = m > mdeg ? m : mdeg;
        }
        if (fallback || active == 0) break;
        // coefficients of the scaled three-term recurrence (Zhou & Saad 2007): damped interval [0, c], sigma_1 = e / (theta_1 - e)
        for (int i = 0; i < mdeg; ++i) { step[i].active = active; step[i].dev = nullptr; step[i].dev_step = 0; }
        for (int z = 0; z < batch; ++z) {
            if (!((active >> z) & 1u)) continue;
            const double* th = ws.h_out + n_part + (size_t)z * (b + 4);
            double c = th[b - 1];
            if (!(c > 1e-12 * th[0])) c = 1e-12 * th[0];
            const double e = 0.5 * c, ctr = 0.5 * c, sigma1 = e / (th[0] - ctr);
            double sigma = sigma1;
            for (int i = 0; i < mdeg; ++i) {
                if (i >= deg[z]) { step[i].a[z] = 0.0; step[i].b[z] = 1.0; step[i].g[z] = 0.0; continue; }
                if (i == 0) { step[i].a[z] = sigma1 / e; step[i].b[z] = -ctr * sigma1 / e; step[i].g[z] = 0.0; continue; }
                const double sn = 1.0 / (2.0 / sigma1 - sigma);
                step[i].a[z] = 2.0 * sn / e;
                step[i].b[z] = -ctr * 2.0 * sn / e;
                step[i].g[z] = -sigma * sn;
                sigma = sn;
            }
        }
        // step 1 (Y1 = (sigma_1 / e) (C X - e X)) was written by lead_rot_kernel from the known product
        int prev = ix, cur = iy, fre = ic;
        for (int i = 1; i < mdeg; ++i) {
            mult(C, ws.P[cur], ws.P[prev], ws.P[fre], ne, b, ys, step[i]);
            ++total_mv;
            const int t = prev; prev = cur; cur = fre; fre = t;
        }
        ic = cur; ix = prev; iy = fre;
    }
    LCHK(hipGetLastError());
    if (dbg)
        fprintf(stderr, "[apv lead] n=%d b=%d rank=%d batch=%d: %d passes, %d block products, %.3f ms%s\n", n, b, rank, batch, pass + 1, total_mv,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), fallback ? " -> full solve" : "");
    if (fallback) return APV_OK;
    // U[:, :b] = W^T X, one launch per buffer that holds finished blocks
    for (int buf = 0; buf < 3; ++buf) {
        LeadCoef cf = one;
        cf.active = 0;
        for (int z = 0; z < batch; ++z)
            if (final_buf[z] == buf) cf.active |= 1u << z;
        if (cf.active) mult(WT, ws.P[buf], nullptr, d_U, n, n, (size_t)n * n, cf);
    }
    hipLaunchKernelGGL(lead_lam_kernel, dim3(batch), dim3(64), 0, st, n, b, ws.theta, d_lam);
    LCHK(hipGetLastError());
#undef LCHK
    *done = 1;
    return APV_OK;
}
