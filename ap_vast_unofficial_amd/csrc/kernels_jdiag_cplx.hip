// jdiag for complex Hermitian pairs beyond the per-bin orders (64 < n <= 1024): reference Python/apvast.py:20-36 takes any n.
//
// The per-bin kernels stop at order 64 (the loudspeaker count); the broadband solver (kernels_gevd_large.hip) is real
// symmetric.  A Hermitian pair (A, B) = (Ar + i Ai, Br + i Bi) is the real symmetric pair of order 2n
//
//     A~ = [ Ar  -Ai ]      B~ = [ Br  -Bi ]
//          [ Ai   Ar ]           [ Bi   Br ]
//
// with every eigenvalue twice: z = (u; v) is an eigenvector of the real pair exactly when x = u + i v is one of the complex
// pair, and its partner J z = (-v; u) stands for i x.  z^T B~ z = x^H B x, so the real solver's normalisation is jdiag's.
// Steps, all on the device:
//   1. embed_kernel          the two real matrices of order 2n
//   2. apv_gevd_large        Cholesky, whitening, block Jacobi, back-transform, descending order (kernels_gevd_large.hip)
//   3. candidates_kernel     the 2n real eigenvectors read as complex vectors x_k, one contiguous row each
//   4. loaded_product_kernel y_k = (B + reg I) x_k
//   5. select_kernel         n of the 2n candidates that are independent over C.  With simple eigenvalues that is every
//                            second one (the other is +-i times its neighbour).  Inside a cluster of eigenvalues the solver returns
//                            an arbitrary real basis of a space of real dimension 2m, whose 2m complex readings span only m
//                            complex dimensions; the kernel walks the candidates in order and keeps one when what is left of it
//                            after projecting out the kept vectors of nearby eigenvalues (B-inner product, two passes) still has
//                            a squared B-norm above 0.05, normalising that remainder.  A complex direction y of the cluster
//                            not yet covered has sum_k |<y, x_k>|^2 = 2 over the cluster's candidates, and every rejected
//                            candidate carries less than 0.05 of it: the walk cannot end short for clusters of fewer than 39
//                            complex dimensions (it reports status 3 if it does).
#include <memory>
#include <string>

#include "apv_internal.h"

#include "gevd16_common.h"

namespace {

using C128 = Cx<double>;
constexpr double kKeep = 0.05;          // squared B-norm a candidate's remainder needs to be kept
constexpr double kWindow = 1e-5;        // eigenvalues closer than this (relative to the largest) count as one cluster

// dst [batch][2n][2n] f64  <-  src [batch][n][n] c128
__global__ void embed_kernel(int n, const C128* __restrict__ src, double* __restrict__ dst) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= n) return;
    const size_t n2 = 2 * (size_t)n;
    const C128 a = src[((size_t)blockIdx.z * n + i) * n + j];
    double* d = dst + (size_t)blockIdx.z * n2 * n2;
    d[(size_t)i * n2 + j] = a.x;
    d[(size_t)i * n2 + n + j] = -a.y;
    d[(size_t)(n + i) * n2 + j] = a.y;
    d[(size_t)(n + i) * n2 + n + j] = a.x;
}

// Xc[k][i] = Z[i][k] + i Z[n + i][k]: 32 x 32 tiles through LDS so that both sides are read and written along rows
__global__ void candidates_kernel(int n, const double* __restrict__ Z, C128* __restrict__ Xc) {
    __shared__ double tr[32][33], ti[32][33];
    const size_t n2 = 2 * (size_t)n;
    const double* Zb = Z + (size_t)blockIdx.z * n2 * n2;
    C128* Xb = Xc + (size_t)blockIdx.z * n2 * n;
    const int k0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
    for (int q = threadIdx.y; q < 32; q += blockDim.y) {
        const int i = i0 + q, k = k0 + threadIdx.x;
        const bool in = i < n && k < (int)n2;
        tr[q][threadIdx.x] = in ? Zb[(size_t)i * n2 + k] : 0.0;
        ti[q][threadIdx.x] = in ? Zb[(size_t)(n + i) * n2 + k] : 0.0;
    }
    __syncthreads();
    for (int q = threadIdx.y; q < 32; q += blockDim.y) {
        const int k = k0 + q, i = i0 + threadIdx.x;
        if (k < (int)n2 && i < n) Xb[(size_t)k * n + i] = mk<double>(tr[threadIdx.x][q], ti[threadIdx.x][q]);
    }
}

// Yc[k][i] = sum_j B[i][j] Xc[k][j] + reg Xc[k][i]; B is Hermitian, so B[i][j] = conj(B[j][i]) is read along row j.
// One workgroup: 64 values of i for 4 candidates k.
__global__ void __launch_bounds__(256) loaded_product_kernel(int n, const C128* __restrict__ B, const C128* __restrict__ Xc, double reg,
                                                             const double* __restrict__ reg_scale, C128* __restrict__ Yc) {
    const size_t n2 = 2 * (size_t)n;
    const C128* Bb = B + (size_t)blockIdx.z * n * n;
    const C128* Xb = Xc + (size_t)blockIdx.z * n2 * n;
    C128* Yb = Yc + (size_t)blockIdx.z * n2 * n;
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int k = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (k >= (int)n2) return;
    const double load = reg_scale ? reg * reg_scale[blockIdx.z] : reg;
    double sx = 0, sy = 0;
    if (i < n) {
        const C128* xk = Xb + (size_t)k * n;
        for (int j = 0; j < n; ++j) {
            const C128 b = Bb[(size_t)j * n + i], x = xk[j];             // conj(b) * x
            sx = fma_t(b.y, x.y, fma_t(b.x, x.x, sx));
            sy = fma_t(-b.y, x.x, fma_t(b.x, x.y, sy));
        }
        const C128 x = xk[i];
        Yb[(size_t)k * n + i] = mk<double>(sx + load * x.x, sy + load * x.y);
    }
}

// one workgroup of 1024 threads per matrix; thread i owns element i of every vector (n <= 1024).
// Q, BQ: [n][n] kept vectors and their products with the loaded B, one row each.  U[i][j] = Q[j][i] at the end.
__global__ void __launch_bounds__(1024) select_kernel(int n, const C128* __restrict__ Xc, const C128* __restrict__ Yc,
                                                      const double* __restrict__ lam2, C128* __restrict__ Q, C128* __restrict__ BQ,
                                                      C128* __restrict__ U, double* __restrict__ lam, int32_t* __restrict__ status) {
    __shared__ C128 s_r[1024], s_c[1024];
    __shared__ double s_lam[1024], s_red[16];
    const size_t n2 = 2 * (size_t)n;
    const int z = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const C128* Xb = Xc + (size_t)z * n2 * n;
    const C128* Yb = Yc + (size_t)z * n2 * n;
    const double* l2 = lam2 + (size_t)z * n2;
    C128* Qb = Q + (size_t)z * n * n;
    C128* BQb = BQ + (size_t)z * n * n;
    const bool own = tid < n;
    const double window = kWindow * fmax(fabs(l2[0]), fabs(l2[n2 - 1]));
    int m = 0;
    for (int k = 0; k < (int)n2 && m < n; ++k) {
        C128 r = own ? Xb[(size_t)k * n + tid] : mk<double>(0, 0);
        C128 br = own ? Yb[(size_t)k * n + tid] : mk<double>(0, 0);
        const double lk = l2[k];
        int j0 = m;                                             // kept vectors j0 .. m-1 share the candidate's cluster
        while (j0 > 0 && s_lam[j0 - 1] - lk <= window) --j0;
        const int cnt = m - j0;
        for (int pass = 0; pass < 2 && cnt > 0; ++pass) {
            s_r[tid] = r;
            __syncthreads();
            for (int q = wave; q < cnt; q += 16) {              // c_q = (B q_j)^H r
                const C128* bq = BQb + (size_t)(j0 + q) * n;
                double cx = 0, cy = 0;
                for (int i = lane; i < n; i += 64) {
                    const C128 b = bq[i], v = s_r[i];
                    cx = fma_t(b.y, v.y, fma_t(b.x, v.x, cx));
                    cy = fma_t(-b.y, v.x, fma_t(b.x, v.y, cy));
                }
                cx = wave_sum(cx);
                cy = wave_sum(cy);
                if (lane == 0) s_c[q] = mk<double>(cx, cy);
            }
            __syncthreads();
            if (own) {
                for (int q = 0; q < cnt; ++q) {
                    const C128 c = s_c[q], qv = Qb[(size_t)(j0 + q) * n + tid], bv = BQb[(size_t)(j0 + q) * n + tid];
                    r.x = fma_t(c.y, qv.y, fma_t(-c.x, qv.x, r.x));
                    r.y = fma_t(-c.y, qv.x, fma_t(-c.x, qv.y, r.y));
                    br.x = fma_t(c.y, bv.y, fma_t(-c.x, bv.x, br.x));
                    br.y = fma_t(-c.y, bv.x, fma_t(-c.x, bv.y, br.y));
                }
            }
            __syncthreads();                                    // s_r, s_c are rewritten by the next pass / candidate
        }
        double rho = wave_sum(r.x * br.x + r.y * br.y);         // r^H (B r), real
        if (lane == 0) s_red[wave] = rho;
        __syncthreads();
        rho = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) rho += s_red[w];
        if (rho > kKeep) {                                      // uniform
            const double inv = rsq_full(rho);
            if (own) {
                Qb[(size_t)m * n + tid] = mk<double>(r.x * inv, r.y * inv);
                BQb[(size_t)m * n + tid] = mk<double>(br.x * inv, br.y * inv);
            }
            if (tid == 0) s_lam[m] = lk;
            ++m;
        }
        __threadfence_block();
        __syncthreads();                                        // the kept rows are read through global memory by every wave
    }
    if (tid == 0 && status != nullptr && status[z] == 0 && m < n) status[z] = 3;
    // outputs: U[i][j] = Q[j][i], eigenvalues of the kept candidates (descending already)
    C128* Ub = U + (size_t)z * n * n;
    for (int j = 0; j < n; ++j)
        if (own) Ub[(size_t)tid * n + j] = j < m ? Qb[(size_t)j * n + tid] : mk<double>(0, 0);
    if (own) lam[(size_t)z * n + tid] = tid < m ? s_lam[tid] : 0.0;
}

struct DevBufs {
    void* p[12] = {};
    ~DevBufs() { for (void* q : p) if (q) (void)hipFree(q); }
};

}  // namespace

int apv_jdiag_large_c128(apv_handle* h, int32_t n, int32_t batch, const void* h_A, const void* h_B, void* h_U, double* h_lam,
                         int32_t* h_status) {
    if (!h || !h_A || !h_B || !h_U || !h_lam) return apv_fail(h, APV_ERR_ARG, "null host pointer");
    if (n < 1 || n > 1024 || batch < 0) return apv_fail(h, APV_ERR_ARG, "apv_jdiag_large_c128: n must be in 1..1024");
    if (batch == 0) return APV_OK;
#define JCHK(call)                                                                                                  \
    do {                                                                                                            \
        hipError_t _e = (call);                                                                                     \
        if (_e != hipSuccess) return apv_fail(h, APV_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e));   \
    } while (0)
    JCHK(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    const size_t n2 = 2 * (size_t)n;
    const size_t cmat = (size_t)batch * n * n * sizeof(C128), rmat = (size_t)batch * n2 * n2 * sizeof(double);
    const size_t cand = (size_t)batch * n2 * n * sizeof(C128);
    DevBufs t;                                                  // freed on every way out
    C128 *dA, *dB, *dXc, *dYc, *dQ, *dBQ, *dU;
    double *dAe, *dBe, *dZ, *dl2, *dl, *dn = nullptr;
    int32_t* dst;
    JCHK(hipMalloc(&t.p[0], cmat)); dA = (C128*)t.p[0];
    JCHK(hipMalloc(&t.p[1], cmat)); dB = (C128*)t.p[1];
    JCHK(hipMalloc(&t.p[2], rmat)); dAe = (double*)t.p[2];
    JCHK(hipMalloc(&t.p[3], rmat)); dBe = (double*)t.p[3];
    JCHK(hipMalloc(&t.p[4], rmat)); dZ = (double*)t.p[4];
    JCHK(hipMalloc(&t.p[5], (size_t)batch * n2 * sizeof(double))); dl2 = (double*)t.p[5];
    JCHK(hipMalloc(&t.p[6], cand)); dXc = (C128*)t.p[6];
    JCHK(hipMalloc(&t.p[7], cand)); dYc = (C128*)t.p[7];
    JCHK(hipMalloc(&t.p[8], cmat)); dQ = (C128*)t.p[8];
    JCHK(hipMalloc(&t.p[9], cmat)); dBQ = (C128*)t.p[9];
    JCHK(hipMalloc(&t.p[10], (size_t)batch * (n * sizeof(double) + sizeof(int32_t)))); dl = (double*)t.p[10];
    dst = (int32_t*)(dl + (size_t)batch * n);
    dU = dA;                                                    // A is spent once it has been embedded
    JCHK(hipMemcpyAsync(dA, h_A, cmat, hipMemcpyHostToDevice, st));
    JCHK(hipMemcpyAsync(dB, h_B, cmat, hipMemcpyHostToDevice, st));
    const dim3 eg((n + 255) / 256, n, batch);
    hipLaunchKernelGGL(embed_kernel, eg, dim3(256), 0, st, n, dA, dAe);
    hipLaunchKernelGGL(embed_kernel, eg, dim3(256), 0, st, n, dB, dBe);
    std::unique_ptr<int32_t[]> tmp;
    int32_t* stat = h_status;
    if (!stat) {
        tmp.reset(new int32_t[batch]);
        stat = tmp.get();
    }
    if (h->cfg.reg_mode == APV_REG_REL) {
        // B + reg_dark ||B||_2 I (apvast.py:26-27); the embedding has B's spectrum
        JCHK(hipMalloc(&t.p[11], (size_t)batch * sizeof(double)));
        dn = (double*)t.p[11];
        for (int z0 = 0; z0 < batch; z0 += 4) {
            const double* mats[4];
            const int cnt = batch - z0 < 4 ? batch - z0 : 4;
            for (int q = 0; q < cnt; ++q) mats[q] = dBe + (size_t)(z0 + q) * n2 * n2;
            JCHK(apv_launch_norm2((int)n2, cnt, mats, dn + z0, st));
        }
    }
    // The real solver stops when a sweep's pivots weigh 1e-16 ||C||^2, which pins eigenVALUES and leaves eigenvectors good to
    // ~1e-8 at order 1000 -- any orthonormal basis satisfies the real contract.  Here a vector's partner J z must lie in the same
    // eigenspace, which takes accurate vectors: one more pair of sweeps.
    h->gl_tol2 = 1e-26;
    const int rc = apv_gevd_large(h, (int)n2, batch, dAe, dBe, h->cfg.reg_dark, dn, dZ, dl2, nullptr, 0.0, 0, nullptr, nullptr, stat);
    h->gl_tol2 = 0.0;
    if (rc != APV_OK && rc != APV_ERR_NOT_PD) return rc;
    if (rc == APV_OK) {
        JCHK(hipMemcpyAsync(dst, stat, (size_t)batch * sizeof(int32_t), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(candidates_kernel, dim3((unsigned)((n2 + 31) / 32), (n + 31) / 32, batch), dim3(32, 8), 0, st, n, dZ, dXc);
        hipLaunchKernelGGL(loaded_product_kernel, dim3((n + 63) / 64, (unsigned)((n2 + 3) / 4), batch), dim3(256), 0, st, n, dB, dXc,
                           h->cfg.reg_dark, dn, dYc);
        hipLaunchKernelGGL(select_kernel, dim3(batch), dim3(1024), 0, st, n, dXc, dYc, dl2, dQ, dBQ, dU, dl, dst);
        JCHK(hipGetLastError());
        JCHK(hipMemcpyAsync(h_U, dU, cmat, hipMemcpyDeviceToHost, st));
        JCHK(hipMemcpyAsync(h_lam, dl, (size_t)batch * n * sizeof(double), hipMemcpyDeviceToHost, st));
        JCHK(hipMemcpyAsync(stat, dst, (size_t)batch * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    }
    JCHK(hipStreamSynchronize(st));
#undef JCHK
    if (rc != APV_OK) return rc;
    for (int z = 0; z < batch; ++z) {
        if (stat[z] == 2) return apv_fail(h, APV_ERR_NO_CONVERGE, "eigen-iteration did not converge");
        if (stat[z] == 3) return apv_fail(h, APV_ERR_NO_CONVERGE, "complex jdiag: an eigenvalue cluster too large to separate");
    }
    return APV_OK;
}
