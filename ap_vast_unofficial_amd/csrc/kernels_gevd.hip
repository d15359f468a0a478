// Batched joint diagonalisation (GEVD) + variable-span filter, LDS-resident.
//
// One workgroup owns one problem (one frequency bin): the bright/dark pair
// (A, B) of order n <= NMAX lives in LDS from the first load to the last
// store; nothing but the inputs and the filter touch HBM.
//
// Stages (reference Python/apvast.py, spec Matlab/ControlMethods/jdiag.m:103-117):
//   0  [fused]  R_B = X_B^H X_B, R_D = X_D^H X_D, r = X_B^H d      apvast.py:329-364 per bin
//   1  Bc = chol(B + reg I)                                        apvast.py:22-27
//   2  C  = Bc^-1 A Bc^-H   (two forward substitutions)            apvast.py:28-29
//   3  C  = Q diag(lam) Q^H (cyclic Jacobi, round-robin ordering)  apvast.py:30  (schur of a Hermitian C)
//   4  sort lam descending                                         apvast.py:32-35
//   5  X  = Bc^-H Q         (backward substitution)                apvast.py:31
//   6  w_V = sum_{i<V} (x_i^H r)/(lam_i+mu) x_i                    apvast.py:406-414
//
// gfx950 only.  Wavefront = 64.
#include "apv_internal.h"

#include <cstdlib>

namespace {

template <typename T>
struct Cx {
    T x, y;
};

template <typename T> __device__ __forceinline__ Cx<T> mk(T a, T b) { Cx<T> r; r.x = a; r.y = b; return r; }
template <typename T> __device__ __forceinline__ Cx<T> cadd(Cx<T> a, Cx<T> b) { return mk<T>(a.x + b.x, a.y + b.y); }
template <typename T> __device__ __forceinline__ Cx<T> csub(Cx<T> a, Cx<T> b) { return mk<T>(a.x - b.x, a.y - b.y); }
template <typename T> __device__ __forceinline__ Cx<T> cmul(Cx<T> a, Cx<T> b) {
    return mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * conj(b)
template <typename T> __device__ __forceinline__ Cx<T> cmulc(Cx<T> a, Cx<T> b) {
    return mk<T>(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
// conj(a) * b
template <typename T> __device__ __forceinline__ Cx<T> ccmul(Cx<T> a, Cx<T> b) {
    return mk<T>(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
}
template <typename T> __device__ __forceinline__ Cx<T> cscale(Cx<T> a, T s) { return mk<T>(a.x * s, a.y * s); }
template <typename T> __device__ __forceinline__ Cx<T> cconj(Cx<T> a) { return mk<T>(a.x, -a.y); }
template <typename T> __device__ __forceinline__ T cabs2(Cx<T> a) { return a.x * a.x + a.y * a.y; }

template <typename T> struct Tol;
template <> struct Tol<double> {
    // quadratic convergence: the sweep that meets it leaves ~tol2^2 behind.  The threshold is relative to ||C||_F, so the
    // vectors of eigenvalues far below ||C|| need it tight: see Prec<double> in gevd16_common.h for the measurements
    static constexpr double sweep_tol2 = 1e-16;
    static constexpr int max_sweeps = 16;
};
template <> struct Tol<float> {
    static constexpr float sweep_tol2 = 1e-8f;
    static constexpr int max_sweeps = 14;
};

// 1/sqrt(x) to the full precision of T (v_rsq_f64 is good to 5e-8 on gfx950: one third-order correction)
__device__ __forceinline__ double rsq_full(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
}
__device__ __forceinline__ float rsq_full(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float e = __builtin_fmaf(-(x * y), y, 1.0f);
    return __builtin_fmaf(y * e, 0.5f, y);
}

// Jacobi rotation J = [[c, s], [-conj(s), c]] for the Hermitian 2x2 [[alpha, beta], [conj(beta), gamma]].
// The angle t = sign(tau) e^{i arg beta} / (|tau| + sqrt(1 + tau^2)), tau = (gamma - alpha) / (2 |beta|), is
// evaluated in float on operands brought to [0.5, 1) by a common power of two (the double-precision divides and
// square roots sat on the serial path of every round); c = 1/sqrt(1 + |t|^2), s = t c are then formed in T, so J
// is unitary to the precision of T and the pivot drops by ~1e-7 instead of to zero -- the next sweep finishes it.
// A pivot below 1e-19 of the larger of (|gamma - alpha|, |beta|) underflows in float and is left alone.
template <typename T>
__device__ __forceinline__ void rotation_scaled(T alpha, T gamma, T bx, T by, T& c, T& sx, T& sy) {
    const T d = gamma - alpha;
    const T m = fmax(fabs(d), fmax(fabs(bx), fabs(by)));
    float fd, fbx, fby;
    if constexpr (sizeof(T) == 8) {
        const int ex = -__builtin_amdgcn_frexp_exp(m);
        fd = (float)__builtin_ldexp(d, ex);
        fbx = (float)__builtin_ldexp(bx, ex);
        fby = (float)__builtin_ldexp(by, ex);
    } else {
        const int ex = -__builtin_amdgcn_frexp_expf(m);
        fd = __builtin_ldexpf(d, ex);
        fbx = __builtin_ldexpf(bx, ex);
        fby = __builtin_ldexpf(by, ex);
    }
    const float b2 = fbx * fbx + fby * fby;
    float tx = 0.f, ty = 0.f;
    if (b2 > 1e-37f) {
        const float iab = __builtin_amdgcn_rsqf(b2);
        const float tau = fd * 0.5f * iab;
        const float h2 = __builtin_fmaf(tau, tau, 1.0f);
        const float rho = (h2 < 3e38f) ? h2 * __builtin_amdgcn_rsqf(h2) : fabsf(tau);
        const float t = copysignf(__builtin_amdgcn_rcpf(fabsf(tau) + rho), tau) * iab;
        tx = fbx * t;
        ty = fby * t;
    }
    const T dx = (T)tx, dy = (T)ty;
    c = rsq_full((T)1 + dx * dx + dy * dy);
    sx = dx * c;
    sy = dy * c;
}

// Round-robin (tournament) pairing: ne players (even), round r in [0, ne-1), slot a in [0, ne/2).
__device__ __forceinline__ void rr_pair(int ne, int r, int a, int& p, int& q) {
    int m1 = ne - 1;
    int u, v;
    if (a == 0) {
        u = m1;
        v = r;
    } else {
        u = (r + a) % m1;
        v = (r - a + m1) % m1;
    }
    p = u < v ? u : v;
    q = u < v ? v : u;
}

// XT: element type of the fused input slabs (float2 = c64, double2 = c128: the float64 streaming front-end)
template <typename T, int NMAX, int TPB, bool FUSED, bool SPILL, bool EXACT, typename XT>
__global__ void __launch_bounds__(TPB) gevd_vast_kernel(const GevdParams p) {
    // zone program of a two-zone launch (blockIdx.y); the argument block itself stays in scalar registers
    const bool z1 = (blockIdx.y == 1);
    const XT* const pXB = reinterpret_cast<const XT*>(z1 ? p.XB1 : p.XB);
    const XT* const pXD = reinterpret_cast<const XT*>(z1 ? p.XD1 : p.XD);
    const XT* const pd = reinterpret_cast<const XT*>(z1 ? p.d1 : p.d);
    void* const pw = z1 ? p.w1 : p.w;
    void* const plam = z1 ? p.lam1 : p.lam;
    int32_t* const pstatus = z1 ? p.status1 : p.status;
    using C = Cx<T>;
    constexpr int LD = NMAX + 1;
    constexpr int NP = NMAX / 2;
    constexpr int MT = 16;                                  // control-point rows staged per step (fused)
    constexpr int NACC = (NMAX * NMAX + TPB - 1) / TPB;

    __shared__ C sA[NMAX * LD];
    __shared__ C sB[NMAX * LD];
    __shared__ C sVstore[SPILL ? 1 : NMAX * LD];
    __shared__ XT sX[FUSED ? MT * NMAX : 1];
    __shared__ XT sd[FUSED ? MT : 1];
    __shared__ C sr[NMAX];
    __shared__ C scoef[NMAX];
    __shared__ T sDiag[NMAX];
    __shared__ T sDinv[NMAX];
    __shared__ T sLam[NMAX];
    __shared__ T sRow[NMAX];
    __shared__ int sOrder[NMAX];
    __shared__ int sMs[64];                                   // multisection votes of the spectral-norm estimate
    __shared__ T rc[NP];
    __shared__ C rs[NP];
    __shared__ T roff[NP];
    __shared__ int rp[NP];
    __shared__ int rq[NP];

    C* sV = SPILL ? sB : sVstore;

    const int n = EXACT ? NMAX : p.n;          // EXACT: order known at compile time (index arithmetic folds)
    const int tid = threadIdx.x;
    const int k = blockIdx.x;
    int status = 0;

    // ---------------- stage 0: load or correlate ----------------
    if constexpr (FUSED) {
        const int M = p.M;
        // float64 with enough waves: the contraction runs on v_mfma_f64_16x16x4_f64, one 16 x 16 tile of R per wave
        // (re and im accumulators), operands read from the staged rows; otherwise one element of R per thread on the VALU
        constexpr bool MFMA_CORR = sizeof(T) == 8 && NMAX >= 16 && (NMAX / 16) * (NMAX / 16) <= TPB / 64;
        for (int which = 0; which < 2; ++which) {
            const XT* X = (which ? pXD : pXB) + (size_t)k * M * n;
            C acc[NACC];
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = mk<T>(0, 0);
            using d4v = __attribute__((ext_vector_type(4))) double;
            d4v mre = {0, 0, 0, 0}, mim = {0, 0, 0, 0};
            const int wave = tid >> 6, lane = tid & 63, il = lane & 15, kq = lane >> 4;
            const int ntile = (n + 15) >> 4;
            const int ti = wave / ntile, tj = wave - ti * ntile;
            const bool tile_ok = wave < ntile * ntile;
            C racc = mk<T>(0, 0);
            for (int m0 = 0; m0 < M; m0 += MT) {
                const int rows = (M - m0) < MT ? (M - m0) : MT;
                for (int idx = tid; idx < rows * n; idx += TPB) sX[idx] = X[(size_t)m0 * n + idx];
                if (which == 0)
                    for (int idx = tid; idx < rows; idx += TPB) sd[idx] = pd[(size_t)k * M + m0 + idx];
                __syncthreads();
                if constexpr (MFMA_CORR) {
                    if (tile_ok) {
                        const int ia = 16 * ti + il, jb = 16 * tj + il;
#pragma unroll
                        for (int m = 0; m < MT; m += 4) {
                            const bool ok = m + kq < rows;
                            XT xzero;
                            xzero.x = 0;
                            xzero.y = 0;
                            const XT xa = (ok && ia < n) ? sX[(m + kq) * n + ia] : xzero;
                            const XT xb = (ok && jb < n) ? sX[(m + kq) * n + jb] : xzero;
                            const double ar = xa.x, ai = xa.y, br = xb.x, bi = xb.y;
                            // conj(x_i) x_j: re = ar br + ai bi, im = ar bi - ai br
                            mre = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, mre, 0, 0, 0);
                            mre = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, bi, mre, 0, 0, 0);
                            mim = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, mim, 0, 0, 0);
                            mim = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, br, mim, 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int a = 0; a < NACC; ++a) {
                        const int idx = tid + a * TPB;
                        if (idx < n * n) {
                            const int i = idx / n, j = idx - i * n;
                            C s = acc[a];
                            for (int m = 0; m < rows; ++m) {
                                const XT xi = sX[m * n + i], xj = sX[m * n + j];
                                // conj(xi) * xj, products exact in T=double
                                s.x += (T)xi.x * (T)xj.x + (T)xi.y * (T)xj.y;
                                s.y += (T)xi.x * (T)xj.y - (T)xi.y * (T)xj.x;
                            }
                            acc[a] = s;
                        }
                    }
                }
                if (which == 0 && tid < n) {
                    for (int m = 0; m < rows; ++m) {
                        const XT xi = sX[m * n + tid], dm = sd[m];
                        racc.x += (T)xi.x * (T)dm.x + (T)xi.y * (T)dm.y;
                        racc.y += (T)xi.x * (T)dm.y - (T)xi.y * (T)dm.x;
                    }
                }
                __syncthreads();
            }
            C* dst = which ? sB : sA;
            if constexpr (MFMA_CORR) {
                // f64 16x16x4 accumulator: row = (lane >> 4) + 4 t, col = lane & 15
                if (tile_ok) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int i = 16 * ti + kq + 4 * t, j = 16 * tj + il;
                        if (i < n && j < n) dst[i * LD + j] = mk<T>((T)mre[t], (T)mim[t]);
                    }
                }
            } else {
#pragma unroll
                for (int a = 0; a < NACC; ++a) {
                    const int idx = tid + a * TPB;
                    if (idx < n * n) {
                        const int i = idx / n, j = idx - i * n;
                        dst[i * LD + j] = acc[a];
                    }
                }
            }
            if (which == 0 && tid < n) sr[tid] = racc;
        }
    } else {
        const C* RB = reinterpret_cast<const C*>(p.RB) + (size_t)k * n * n;
        const C* RD = reinterpret_cast<const C*>(p.RD) + (size_t)k * n * n;
        for (int idx = tid; idx < n * n; idx += TPB) {
            const int i = idx / n, j = idx - i * n;
            sA[i * LD + j] = RB[idx];
            sB[i * LD + j] = RD[idx];
        }
        if (p.r != nullptr) {
            if (tid < n) sr[tid] = reinterpret_cast<const C*>(p.r)[(size_t)k * n + tid];
        } else if (tid < n) {
            sr[tid] = mk<T>(0, 0);
        }
    }
    __syncthreads();
    if (p.debug_stop == 1) return;

    // ---------------- stage 1: loading + Cholesky of B ----------------
    T load = (T)p.reg_dark;
    T load_bright = 0;
    if (p.reg_mode == APV_REG_REL || p.reg_bright != 0.0) {
        // spectral norm (||.||_2 of apvast.py:26, apVast.m:552-569) = largest eigenvalue of the Hermitian PSD matrix: n steps of
        // the Lanczos recurrence (no re-orthogonalisation: the largest Ritz value does not need it) from a start vector with no
        // symmetry (an all-ones start is orthogonal to the dominant eigenvector of many array geometries), then the largest
        // eigenvalue of the tridiagonal matrix by 64-way multisection on Sturm counts.  sLam / sDiag hold alpha / beta.
        for (int which = 0; which < 2; ++which) {
            const bool need = which ? (p.reg_mode == APV_REG_REL) : (p.reg_bright != 0.0);
            if (!need) continue;
            const C* Mx = which ? sB : sA;
            C vcur = mk<T>(0, 0), vprev = mk<T>(0, 0);
            {
                const T f = (T)tid * (T)0.6180339887498949;
                if (tid < n) vcur = mk<T>((T)1 + (f - floor(f)), (T)0);
                if (tid < n) sRow[tid] = cabs2(vcur);
                __syncthreads();
                T s2 = 0;
                for (int j = 0; j < n; ++j) s2 += sRow[j];
                vcur = cscale(vcur, (T)1 / sqrt(s2));
                __syncthreads();
            }
            T beta_prev = 0;
            int m = 0;
            for (int it = 0; it < n; ++it) {
                if (tid < n) scoef[tid] = vcur;
                __syncthreads();
                C y = mk<T>(0, 0);
                if (tid < n)
                    for (int j = 0; j < n; ++j) y = cadd(y, cmul(Mx[tid * LD + j], scoef[j]));
                if (tid < n) sRow[tid] = vcur.x * y.x + vcur.y * y.y;                  // Re(conj(v) y)
                __syncthreads();
                T alpha = 0;
                for (int j = 0; j < n; ++j) alpha += sRow[j];
                __syncthreads();
                C w = csub(csub(y, cscale(vcur, alpha)), cscale(vprev, beta_prev));
                if (tid < n) sRow[tid] = cabs2(w);
                if (tid == 0) sLam[it] = alpha;
                __syncthreads();
                T b2 = 0;
                for (int j = 0; j < n; ++j) b2 += sRow[j];
                const T beta = sqrt(b2);
                if (tid == 0) sDiag[it] = beta;
                m = it + 1;
                __syncthreads();
                if (!(beta > (T)1e-14 * fabs(alpha)) || it + 1 == n) break;            // invariant subspace reached (uniform)
                vprev = vcur;
                vcur = cscale(w, (T)1 / beta);
                beta_prev = beta;
            }
            // Gershgorin interval of the m x m tridiagonal matrix, then nine passes of 64-way multisection
            T lo = 0, hi = 0;
            for (int i = 0; i < m; ++i) {
                const T r = (i > 0 ? sDiag[i - 1] : (T)0) + (i + 1 < m ? sDiag[i] : (T)0);
                hi = fmax(hi, sLam[i] + r);
            }
            for (int pass = 0; pass < 9 && hi > lo; ++pass) {
                const int t = tid & 63;
                const T x = lo + (hi - lo) * (T)(t + 1) / (T)65;
                int below = 0;                                                         // eigenvalues < x
                // Sturm sequence; a vanishing pivot is nudged to the smallest normal number of T (a literal like 1e-300 is zero
                // in float and the quotient a NaN), and a NaN pivot counts as negative: the count can only err towards "x is
                // above", i.e. towards a LOWER estimate, never towards an inflated norm
                constexpr T kTiny = sizeof(T) == 8 ? (T)2.2250738585072014e-308 : (T)1.17549435e-38f;
                T dq = sLam[0] - x;
                below += !(dq >= (T)0);
                for (int i = 1; i < m; ++i) {
                    if (fabs(dq) < kTiny) dq = dq < (T)0 ? -kTiny : kTiny;
                    dq = sLam[i] - x - sDiag[i - 1] * sDiag[i - 1] / dq;
                    below += !(dq >= (T)0);
                }
                __syncthreads();
                if (tid < 64) sMs[t] = (below < m) ? 1 : 0;                            // x_t still below the largest eigenvalue
                __syncthreads();
                int q = 0;
                for (int j = 0; j < 64; ++j) q += sMs[j];
                const T step = (hi - lo) / (T)65;
                const T nlo = lo + step * (T)q, nhi = lo + step * (T)(q + 1);
                lo = nlo;
                hi = q < 64 ? nhi : hi;
                __syncthreads();
            }
            const T nrm = (T)0.5 * (lo + hi);
            if (which) load = (T)p.reg_dark * nrm; else load_bright = (T)p.reg_bright * nrm;
        }
    }
    if (tid < n) {
        C b = sB[tid * LD + tid];
        sB[tid * LD + tid] = mk<T>(b.x + load, 0);
        C a = sA[tid * LD + tid];
        sA[tid * LD + tid] = mk<T>(a.x + load_bright, 0);
    }
    __syncthreads();

    for (int kk = 0; kk < n; ++kk) {
        const T dkk = sB[kk * LD + kk].x;
        if (!(dkk > (T)0) || !(dkk < (T)3.0e38)) {        // uniform: every thread reads the same word
            status = 1;
            break;
        }
        const T inv = (T)1 / sqrt(dkk);
        if (tid == 0) {
            sDiag[kk] = sqrt(dkk);
            sDinv[kk] = inv;
        }
        for (int i = kk + 1 + tid; i < n; i += TPB) sB[i * LD + kk] = cscale(sB[i * LD + kk], inv);
        __syncthreads();
        const int rem = n - kk - 1;                        // trailing block is rem x rem (lower part used)
        for (int idx = tid; idx < rem * rem; idx += TPB) {
            const int i = kk + 1 + idx / rem, j = kk + 1 + idx % rem;
            if (j <= i) sB[i * LD + j] = csub(sB[i * LD + j], cmulc(sB[i * LD + kk], sB[j * LD + kk]));
        }
        __syncthreads();
    }

    if (p.debug_stop == 2) return;
    if (status == 0) {
        // ---------------- stage 2: C = L^-1 A L^-H ----------------
        for (int pass = 0; pass < 2; ++pass) {
            for (int kk = 0; kk < n; ++kk) {
                const T inv = sDinv[kk];
                for (int j = tid; j < n; j += TPB) sA[kk * LD + j] = cscale(sA[kk * LD + j], inv);
                __syncthreads();
                const int rem = n - kk - 1;
                for (int idx = tid; idx < rem * n; idx += TPB) {
                    const int i = kk + 1 + idx / n, j = idx % n;
                    sA[i * LD + j] = csub(sA[i * LD + j], cmul(sB[i * LD + kk], sA[kk * LD + j]));
                }
                __syncthreads();
            }
            if (pass == 0) {
                // in-place conjugate transpose
                for (int idx = tid; idx < n * n; idx += TPB) {
                    const int i = idx / n, j = idx - i * n;
                    if (i < j) {
                        const C u = sA[i * LD + j], l = sA[j * LD + i];
                        sA[i * LD + j] = cconj(l);
                        sA[j * LD + i] = cconj(u);
                    } else if (i == j) {
                        sA[i * LD + i] = cconj(sA[i * LD + i]);
                    }
                }
                __syncthreads();
            }
        }
        // symmetrise, real diagonal
        for (int idx = tid; idx < n * n; idx += TPB) {
            const int i = idx / n, j = idx - i * n;
            if (i < j) {
                const C u = sA[i * LD + j], l = sA[j * LD + i];
                const C m = mk<T>((T)0.5 * (u.x + l.x), (T)0.5 * (u.y - l.y));
                sA[i * LD + j] = m;
                sA[j * LD + i] = cconj(m);
            } else if (i == j) {
                sA[i * LD + i].y = 0;
            }
        }
        if constexpr (SPILL) {
            // park L in HBM scratch: its LDS slot becomes the eigenvector matrix
            C* Ls = reinterpret_cast<C*>(p.Lspill) + ((size_t)blockIdx.y * p.K + k) * n * n;
            for (int idx = tid; idx < n * n; idx += TPB) {
                const int i = idx / n, j = idx - i * n;
                Ls[idx] = sB[i * LD + j];
            }
        }
        __syncthreads();

        if (p.debug_stop == 3) return;
        // ---------------- stage 3: cyclic Jacobi ----------------
        for (int idx = tid; idx < n * n; idx += TPB) {
            const int i = idx / n, j = idx - i * n;
            sV[i * LD + j] = mk<T>(i == j ? (T)1 : (T)0, (T)0);
        }
        if (n & 1) {
            // odd order: index n is the tournament's bye; give it a zero ghost row and column (n < NMAX)
            for (int j = tid; j <= n; j += TPB) {
                sA[n * LD + j] = mk<T>(0, 0);
                sA[j * LD + n] = mk<T>(0, 0);
                sV[n * LD + j] = mk<T>(0, 0);
                sV[j * LD + n] = mk<T>(0, 0);
            }
        }
        if (tid < n) {
            T s = 0;
            for (int j = 0; j < n; ++j) s += cabs2(sA[tid * LD + j]);
            sRow[tid] = s;
        }
        __syncthreads();
        T normF2 = 0;
        for (int j = 0; j < n; ++j) normF2 += sRow[j];

        const int ne = n + (n & 1);
        const int np = ne / 2;
        const int rounds = ne - 1;
        const int max_sweeps = p.max_sweeps > 0 ? p.max_sweeps : Tol<T>::max_sweeps;
        bool converged = (n == 1);
        for (int sweep = 0; sweep < max_sweeps && !converged; ++sweep) {
            T my_off = 0;                       // pair threads: sum of |pivot|^2 over the sweep
            for (int r = 0; r < rounds; ++r) {
                if (tid < np) {
                    int pp, qq;
                    rr_pair(ne, r, tid, pp, qq);
                    T c = 1;
                    C s = mk<T>(0, 0);
                    if (qq < n) {
                        const T alpha = sA[pp * LD + pp].x, gamma = sA[qq * LD + qq].x;
                        const C beta = sA[pp * LD + qq];
                        my_off += cabs2(beta);
                        rotation_scaled<T>(alpha, gamma, beta.x, beta.y, c, s.x, s.y);
                    }
                    // qq == n (odd n): the bye.  Row/column n of C and V are zero ghosts and the
                    // rotation is the identity, so the real index pp still sees its partners' rotations.
                    rp[tid] = pp;
                    rq[tid] = qq;
                    rc[tid] = c;
                    rs[tid] = s;
                }
                __syncthreads();
                // C <- J^H C J on 2x2 blocks
                for (int idx = tid; idx < np * np; idx += TPB) {
                    const int a = idx / np, b = idx - a * np;
                    const int pa = rp[a], qa = rq[a], pb = rp[b], qb = rq[b];
                    const T ca = rc[a], cb = rc[b];
                    const C sa = rs[a], sb = rs[b];
                    C xpp = sA[pa * LD + pb], xpq = sA[pa * LD + qb];
                    C xqp = sA[qa * LD + pb], xqq = sA[qa * LD + qb];
                    // columns: [x_p, x_q] J_b
                    C ypp = csub(cscale(xpp, cb), cmulc(xpq, sb));
                    C ypq = cadd(cmul(xpp, sb), cscale(xpq, cb));
                    C yqp = csub(cscale(xqp, cb), cmulc(xqq, sb));
                    C yqq = cadd(cmul(xqp, sb), cscale(xqq, cb));
                    // rows: J_a^H [y_p; y_q]
                    C zpp = csub(cscale(ypp, ca), cmul(sa, yqp));
                    C zpq = csub(cscale(ypq, ca), cmul(sa, yqq));
                    C zqp = cadd(ccmul(sa, ypp), cscale(yqp, ca));
                    C zqq = cadd(ccmul(sa, ypq), cscale(yqq, ca));
                    if (a == b) {
                        // the rotation angle is good to float accuracy: the pivot keeps its (tiny) computed value
                        zqp = cconj(zpq);
                        zpp.y = 0;
                        zqq.y = 0;
                    }
                    sA[pa * LD + pb] = zpp;
                    sA[pa * LD + qb] = zpq;
                    sA[qa * LD + pb] = zqp;
                    sA[qa * LD + qb] = zqq;
                }
                // V <- V J
                for (int idx = tid; idx < n * np; idx += TPB) {
                    const int i = idx / np, b = idx - i * np;
                    const int pb = rp[b], qb = rq[b];
                    const T cb = rc[b];
                    const C sb = rs[b];
                    const C vp = sV[i * LD + pb], vq = sV[i * LD + qb];
                    sV[i * LD + pb] = csub(cscale(vp, cb), cmulc(vq, sb));
                    sV[i * LD + qb] = cadd(cmul(vp, sb), cscale(vq, cb));
                }
                __syncthreads();
            }
            if (tid < np) roff[tid] = my_off;
            __syncthreads();
            T off = 0;
            for (int a = 0; a < np; ++a) off += roff[a];
            __syncthreads();
            if (off <= (p.sweep_tol2 > 0.0 ? (T)p.sweep_tol2 : Tol<T>::sweep_tol2) * normF2) converged = true;
        }
        if (!converged) status = 2;

        // ---------------- stage 4: eigenvalues, descending order ----------------
        if (tid < n) sLam[tid] = sA[tid * LD + tid].x;
        __syncthreads();
        if (tid < n) {
            const T li = sLam[tid];
            int rank = 0;
            for (int j = 0; j < n; ++j) {
                const T lj = sLam[j];
                rank += (lj > li) || (lj == li && j < tid);
            }
            sOrder[rank] = tid;
        }
        __syncthreads();

        // ---------------- stage 5: X = L^-H Q ----------------
        C* sL = sB;
        if constexpr (SPILL) {
            sL = sA;                                       // C is spent (lam copied out)
            const C* Ls = reinterpret_cast<const C*>(p.Lspill) + ((size_t)blockIdx.y * p.K + k) * n * n;
            for (int idx = tid; idx < n * n; idx += TPB) {
                const int i = idx / n, j = idx - i * n;
                sA[i * LD + j] = Ls[idx];
            }
            __syncthreads();
        }
        for (int kk = n - 1; kk >= 0; --kk) {
            const T inv = sDinv[kk];
            for (int j = tid; j < n; j += TPB) sV[kk * LD + j] = cscale(sV[kk * LD + j], inv);
            __syncthreads();
            for (int idx = tid; idx < kk * n; idx += TPB) {
                const int i = idx / n, j = idx - i * n;
                // X[i][:] -= conj(L[kk][i]) * X[kk][:]
                sV[i * LD + j] = csub(sV[i * LD + j], ccmul(sL[kk * LD + i], sV[kk * LD + j]));
            }
            __syncthreads();
        }

        // ---------------- stage 6: variable-span filter ----------------
        if (tid < n) {
            C s = mk<T>(0, 0);
            for (int l = 0; l < n; ++l) s = cadd(s, ccmul(sV[l * LD + tid], sr[l]));
            const T den = (T)1 / (sLam[tid] + (T)p.mu);
            scoef[tid] = cscale(s, den);
        }
        __syncthreads();
    }

    // ---------------- outputs ----------------
    if (tid < n) {
        C acc = mk<T>(0, 0);
        int done = 0;
        for (int t = 0; t < p.nV; ++t) {
            const int V = p.ranks[t];
            if (status != 1) {
                for (; done < V; ++done) {
                    const int c = sOrder[done];
                    acc = cadd(acc, cmul(scoef[c], sV[tid * LD + c]));
                }
            }
            const size_t o = ((size_t)k * p.nV + t) * n + tid;
            if (p.out_c128) {
                reinterpret_cast<double2*>(pw)[o] = make_double2((double)acc.x, (double)acc.y);
            } else {
                reinterpret_cast<float2*>(pw)[o] = make_float2((float)acc.x, (float)acc.y);
            }
        }
        if (plam != nullptr) {
            const T lv = (status != 1) ? sLam[sOrder[tid]] : (T)0;
            if (p.out_c128) reinterpret_cast<double*>(plam)[(size_t)k * n + tid] = (double)lv;
            else reinterpret_cast<float*>(plam)[(size_t)k * n + tid] = (float)lv;
        }
    }
    if (p.U != nullptr) {
        C* U = reinterpret_cast<C*>(p.U) + (size_t)k * n * n;
        for (int idx = tid; idx < n * n; idx += TPB) {
            const int i = idx / n, j = idx - i * n;
            U[idx] = (status != 1) ? sV[i * LD + sOrder[j]] : mk<T>(0, 0);
        }
    }
    if (pstatus != nullptr && tid == 0) pstatus[k] = status;
}

template <typename T, int NMAX, int TPB, bool SPILL>
hipError_t launch_t(const GevdParams& p, bool fused, hipStream_t s) {
    if (p.K <= 0) return hipSuccess;
    const dim3 grid(p.K, p.n_zones > 1 ? 2 : 1);
    const bool xd = fused && p.x_c128;
    if (p.n == NMAX) {
        if (xd) hipLaunchKernelGGL((gevd_vast_kernel<T, NMAX, TPB, true, SPILL, true, double2>), grid, dim3(TPB), 0, s, p);
        else if (fused) hipLaunchKernelGGL((gevd_vast_kernel<T, NMAX, TPB, true, SPILL, true, float2>), grid, dim3(TPB), 0, s, p);
        else hipLaunchKernelGGL((gevd_vast_kernel<T, NMAX, TPB, false, SPILL, true, float2>), grid, dim3(TPB), 0, s, p);
    } else {
        if (xd) hipLaunchKernelGGL((gevd_vast_kernel<T, NMAX, TPB, true, SPILL, false, double2>), grid, dim3(TPB), 0, s, p);
        else if (fused) hipLaunchKernelGGL((gevd_vast_kernel<T, NMAX, TPB, true, SPILL, false, float2>), grid, dim3(TPB), 0, s, p);
        else hipLaunchKernelGGL((gevd_vast_kernel<T, NMAX, TPB, false, SPILL, false, float2>), grid, dim3(TPB), 0, s, p);
    }
    return hipGetLastError();
}

}  // namespace

size_t apv_gevd_spill_bytes(int n, int K, int compute_dtype, int reg_mode, double reg_bright, double sweep_tol2, int zones) {
    // Sized by the kernel that can actually run.  Per (zone program, bin): the order-64 kernel parks C, W (c128), a
    // float32 matrix and r (either arithmetic: the fused order-64 update runs that kernel for both); the float64 LDS kernel of
    // orders 33..64 (SPILL instance) parks its Cholesky factor, n x n c128; every other instance keeps everything in LDS.
    const size_t z = zones > 1 ? 2 : 1;
    if (apv_gevd64_eligible(n, reg_mode, reg_bright, sweep_tol2)) return z * K * apv_gevd64_slot_bytes();
    if (n > 32 && compute_dtype == APV_F64) return z * K * (size_t)n * n * 16;
    return 0;
}

// will a fused launch of these parameters go to the kernel that reads grouped spectra?  (the conditions of apv_launch_gevd16m's
// grouped branch and of the dispatch below)
int apv_gevd_reads_groups(const GevdParams& p, int compute_dtype, bool x_c128) {
    static const bool force_generic = (getenv("APV_FORCE_GENERIC") != nullptr);
    static const int env = getenv("APV_SPECTRA_GROUP") ? atoi(getenv("APV_SPECTRA_GROUP")) : 4;      // A/B switch: 1 = bin-major spectra, 4 | 8 bins a group
    const int g = (env == 4 || env == 8) ? env : 1;
    const bool ok = g > 1 && !force_generic && p.n == 16 && p.reg_mode == APV_REG_ABS && p.reg_bright == 0.0 && compute_dtype == APV_F64 && x_c128 &&
           p.debug_stop == 0 && p.stamps == nullptr;
    return ok ? g : 1;
}

hipError_t apv_launch_gevd(const GevdParams& p, int compute_dtype, bool fused, hipStream_t s, std::string* why) {
    const int n = p.n;
    static const bool force_generic = (getenv("APV_FORCE_GENERIC") != nullptr);
    if (p.n_hops > 1 && (force_generic || !apv_gevd16m_takes_hops(p, compute_dtype, fused))) {
        if (why) *why = "several hops per launch (n_hops > 1) are taken by the order-16 kernels on fused slabs only";
        return hipErrorInvalidValue;
    }
    if (!force_generic) {
        const hipError_t e16 = apv_launch_gevd16m(p, compute_dtype, fused, s);
        if (e16 != hipErrorNotSupported) return e16;
        const hipError_t e64 = apv_launch_gevd64(p, compute_dtype, fused, s);
        if (e64 != hipErrorNotSupported) return e64;
    }
    if (n < 1 || n > APV_MAX_N) {
        if (why) *why = "GEVD order n out of range (1..64)";
        return hipErrorInvalidValue;
    }
    if (p.x_group > 1) {
        if (why) *why = "grouped spectra (x_group > 1) are read by the order-16 float64 kernel only";
        return hipErrorInvalidValue;
    }

    if (compute_dtype == APV_F64) {
        if (n <= 8) return launch_t<double, 8, 64, false>(p, fused, s);
        if (n <= 16) return launch_t<double, 16, 64, false>(p, fused, s);
        if (n <= 32) return launch_t<double, 32, 256, false>(p, fused, s);
        return launch_t<double, 64, 1024, true>(p, fused, s);
    } else {
        if (n <= 8) return launch_t<float, 8, 64, false>(p, fused, s);
        if (n <= 16) return launch_t<float, 16, 64, false>(p, fused, s);
        if (n <= 32) return launch_t<float, 32, 256, false>(p, fused, s);
        return launch_t<float, 64, 1024, false>(p, fused, s);
    }
}
