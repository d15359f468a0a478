// Joint diagonalisation of REAL symmetric pairs of broadband order (n = filter_length x loudspeakers: 256 at
// cfg1, 800 with the parameters of make_python_test.m), float64, matrices resident in HBM/L2.
//
//   jdiag(A, B)   reference Python/apvast.py:20-36, called at apvast.py:380, 382 with n = J L
//
//   1  elimination on [B + reg I | I]: after n steps the left block holds the (unscaled) columns of the Cholesky
//      factor and the right block W' with W = diag(1/sqrt d) W' = L^-1                         apvast.py:22-27
//   2  C = W A W^T                     two tiled GEMMs                                         apvast.py:28-29
//   3  C = Q diag(lam) Q^T             cyclic Jacobi, round-robin order, one launch per round,
//                                      matrices ping-pong between two buffers                  apvast.py:30
//   4  rank of every eigenvalue (descending), X = W^T Q, columns gathered in that order         apvast.py:31-35
//   5  (optional) w_v = sum_{i<v} (x_i^T r)/(lam_i + mu) x_i for v = 1..V                       apvast.py:406-414
//
// Launch-bound by construction (n + ~8(n-1) small launches); what it buys is the reference's own
// broadband GEVD on the device, checked against fixture G1.
#include "apv_internal.h"

#include <vector>

namespace {

constexpr int TPB = 256;

// ---- step 1 ---------------------------------------------------------------------------------------
// one elimination step kk on the ne x ne working matrices (row-major, leading dimension ld)
__global__ void __launch_bounds__(TPB) chol_inv_step_kernel(int n, int ld, int kk, double* __restrict__ B,
                                                            double* __restrict__ W, double* __restrict__ dinv,
                                                            int* __restrict__ flag, size_t mat_stride,
                                                            size_t vec_stride) {
    const int z = blockIdx.z;
    B += z * mat_stride;
    W += z * mat_stride;
    dinv += z * vec_stride;
    if (flag[z]) return;
    const double d = B[(size_t)kk * ld + kk];
    if (!(d > 0.0) || !(d < 1e300)) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) flag[z] = 1;
        return;
    }
    const double inv2 = 1.0 / d;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) dinv[kk] = 1.0 / sqrt(d);
    // rows i > kk, all columns j: j in (kk, i] updates B, j <= kk updates W'
    const int i = kk + 1 + blockIdx.y;
    if (i >= n) return;
    const double lik = B[(size_t)i * ld + kk];
    for (int j = blockIdx.x * TPB + threadIdx.x; j <= i; j += gridDim.x * TPB) {
        if (j > kk) B[(size_t)i * ld + j] -= lik * B[(size_t)j * ld + kk] * inv2;
        else W[(size_t)i * ld + j] -= lik * W[(size_t)kk * ld + j] * inv2;
    }
}

__global__ void __launch_bounds__(TPB) add_diag_kernel(int n, int ld, double* __restrict__ B, double reg, size_t mat_stride) {
    B += blockIdx.z * mat_stride;
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i < n) B[(size_t)i * ld + i] += reg;
}

__global__ void __launch_bounds__(TPB) scale_rows_kernel(int n, int ld, double* __restrict__ W,
                                                         const double* __restrict__ dinv, size_t mat_stride,
                                                         size_t vec_stride) {
    const int z = blockIdx.z;
    W += z * mat_stride;
    dinv += z * vec_stride;
    const int i = blockIdx.y;
    const double s = dinv[i];
    for (int j = blockIdx.x * TPB + threadIdx.x; j < n; j += gridDim.x * TPB) W[(size_t)i * ld + j] *= s;
}

// ---- step 2: C = op(A) op(B), 16x16 tiles ---------------------------------------------------------
template <bool TA, bool TB>
__global__ void __launch_bounds__(256) gemm_kernel(int n, int ld, const double* __restrict__ A,
                                                   const double* __restrict__ Bm, double* __restrict__ C,
                                                   size_t mat_stride) {
    __shared__ double sa[16][17], sb[16][17];
    const int z = blockIdx.z;
    A += z * mat_stride;
    Bm += z * mat_stride;
    C += z * mat_stride;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int row = blockIdx.y * 16 + ty, col = blockIdx.x * 16 + tx;
    double acc = 0.0;
    for (int k0 = 0; k0 < n; k0 += 16) {
        const int ka = k0 + tx, kb = k0 + ty;
        sa[ty][tx] = (row < n && ka < n) ? (TA ? A[(size_t)ka * ld + row] : A[(size_t)row * ld + ka]) : 0.0;
        sb[ty][tx] = (kb < n && col < n) ? (TB ? Bm[(size_t)col * ld + kb] : Bm[(size_t)kb * ld + col]) : 0.0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += sa[ty][k] * sb[k][tx];
        __syncthreads();
    }
    if (row < n && col < n) C[(size_t)row * ld + col] = acc;
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

__global__ void __launch_bounds__(TPB) set_identity_kernel(int ne, int ld, double* __restrict__ V, size_t mat_stride) {
    V += blockIdx.z * mat_stride;
    const int i = blockIdx.y;
    for (int j = blockIdx.x * TPB + threadIdx.x; j < ne; j += gridDim.x * TPB) V[(size_t)i * ld + j] = (i == j) ? 1.0 : 0.0;
}

// ---- step 3 ---------------------------------------------------------------------------------------
__device__ __forceinline__ void rr_pair(int ne, int r, int a, int& p, int& q) {
    const int m1 = ne - 1;
    int u, v;
    if (a == 0) { u = m1; v = r; } else { u = (r + a) % m1; v = (r - a + m1) % m1; }
    p = u < v ? u : v;
    q = u < v ? v : u;
}

__device__ __forceinline__ void sym_rotation(double alpha, double gamma, double beta, double& c, double& s) {
    c = 1.0;
    s = 0.0;
    const double b2 = beta * beta;
    if (b2 > 1e-290 && b2 > 1e-60 * (alpha * alpha + gamma * gamma)) {
        const double tau = (gamma - alpha) / (2.0 * beta);
        const double t = copysign(1.0, tau) / (fabs(tau) + sqrt(1.0 + tau * tau));
        c = 1.0 / sqrt(1.0 + t * t);
        s = t * c;
    }
}

// one round: thread (a, b) owns rows {p_a, q_a} x columns {p_b, q_b} of C and rows {2a, 2a+1} x the same
// columns of V; everything is read from `cur` and written to `nxt`, so no ordering is needed inside the launch
__global__ void __launch_bounds__(TPB) jacobi_round_kernel(int n, int ne, int ld, int round,
                                                           const double* __restrict__ Ccur, double* __restrict__ Cnxt,
                                                           const double* __restrict__ Vcur, double* __restrict__ Vnxt,
                                                           double* __restrict__ off, size_t mat_stride) {
    const int z = blockIdx.z;
    Ccur += z * mat_stride; Cnxt += z * mat_stride; Vcur += z * mat_stride; Vnxt += z * mat_stride;
    const int np = ne / 2;
    const int idx = blockIdx.x * TPB + threadIdx.x;
    if (idx >= np * np) return;
    const int a = idx / np, b = idx - a * np;
    int pa, qa, pb, qb;
    rr_pair(ne, round, a, pa, qa);
    rr_pair(ne, round, b, pb, qb);
    double ca, sa, cb, sb;
    // an index >= n is the bye of an odd order: identity rotation, zero ghost row/column
    if (qa < n) sym_rotation(Ccur[(size_t)pa * ld + pa], Ccur[(size_t)qa * ld + qa], Ccur[(size_t)pa * ld + qa], ca, sa);
    else { ca = 1.0; sa = 0.0; }
    if (qb < n) sym_rotation(Ccur[(size_t)pb * ld + pb], Ccur[(size_t)qb * ld + qb], Ccur[(size_t)pb * ld + qb], cb, sb);
    else { cb = 1.0; sb = 0.0; }
    const double xpp = Ccur[(size_t)pa * ld + pb], xpq = Ccur[(size_t)pa * ld + qb];
    const double xqp = Ccur[(size_t)qa * ld + pb], xqq = Ccur[(size_t)qa * ld + qb];
    // columns: [x_p, x_q] J_b, J = [[c, s], [-s, c]]
    const double ypp = cb * xpp - sb * xpq, ypq = sb * xpp + cb * xpq;
    const double yqp = cb * xqp - sb * xqq, yqq = sb * xqp + cb * xqq;
    // rows: J_a^T [y_p; y_q]
    double zpp = ca * ypp - sa * yqp, zpq = ca * ypq - sa * yqq;
    double zqp = sa * ypp + ca * yqp, zqq = sa * ypq + ca * yqq;
    if (a == b) {
        if (qa < n) atomicAdd(off + z, xpq * xpq);
        zpq = 0.0;
        zqp = 0.0;
    }
    Cnxt[(size_t)pa * ld + pb] = zpp;
    Cnxt[(size_t)pa * ld + qb] = zpq;
    Cnxt[(size_t)qa * ld + pb] = zqp;
    Cnxt[(size_t)qa * ld + qb] = zqq;
    const int r0 = 2 * a, r1 = 2 * a + 1;
    const double v0p = Vcur[(size_t)r0 * ld + pb], v0q = Vcur[(size_t)r0 * ld + qb];
    const double v1p = Vcur[(size_t)r1 * ld + pb], v1q = Vcur[(size_t)r1 * ld + qb];
    Vnxt[(size_t)r0 * ld + pb] = cb * v0p - sb * v0q;
    Vnxt[(size_t)r0 * ld + qb] = sb * v0p + cb * v0q;
    Vnxt[(size_t)r1 * ld + pb] = cb * v1p - sb * v1q;
    Vnxt[(size_t)r1 * ld + qb] = sb * v1p + cb * v1q;
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

// ---- step 4 ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) rank_kernel(int n, int ld, const double* __restrict__ C, double* __restrict__ lam,
                                                   int* __restrict__ order, size_t mat_stride, size_t vec_stride) {
    C += blockIdx.z * mat_stride;
    lam += blockIdx.z * (size_t)n;              // dense [batch][n]
    order += blockIdx.z * vec_stride;
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const double li = C[(size_t)i * ld + i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
        const double lj = C[(size_t)j * ld + j];
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
    U += blockIdx.z * out_stride;
    lam += blockIdx.z * (size_t)n;              // dense [batch][n]
    r += blockIdx.z * (size_t)n;
    coef += blockIdx.z * vec_stride;
    const int c = blockIdx.x * TPB + threadIdx.x;
    if (c >= n) return;
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += U[(size_t)i * n + c] * r[i];
    coef[c] = s / (lam[c] + mu);
}

// w[v][i] = sum_{c <= v} coef[c] U[i][c], v = 0..V-1 (the reference keeps every rank, apvast.py:406-414)
__global__ void __launch_bounds__(TPB) vast_prefix_kernel(int n, int V, const double* __restrict__ U,
                                                          const double* __restrict__ coef, double* __restrict__ w,
                                                          size_t out_stride, size_t vec_stride, size_t w_stride) {
    U += blockIdx.z * out_stride;
    coef += blockIdx.z * vec_stride;
    w += blockIdx.z * w_stride;
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
    for (int v = 0; v < V; ++v) {
        acc += coef[v] * U[(size_t)i * n + v];
        w[(size_t)v * n + i] = acc;
    }
}

}  // namespace

// Workspace + captured two-sweep graph, cached on the handle (sizes rarely change between hops).
struct GevdLargeWs {
    int n = 0, batch = 0;
    double *Bw = nullptr, *W = nullptr, *T1 = nullptr, *C0 = nullptr, *C1 = nullptr, *V0 = nullptr, *V1 = nullptr;
    double *dinv = nullptr, *acc = nullptr, *coef = nullptr;
    int *flag = nullptr, *order = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    void release() {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        void* bufs[] = {Bw, W, T1, C0, C1, V0, V1, dinv, acc, coef, flag, order};
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
// with +reg on its diagonal here); outputs d_U [batch][n][n] (sorted columns), d_lam [batch][n]; optional
// d_r [batch][n] -> d_w [batch][V][n].  h_status[batch]: 0 ok, 1 not positive definite, 2 sweep cap.
int apv_gevd_large(apv_handle* h, int n, int batch, const double* d_A, const double* d_B, double reg, double* d_U,
                   double* d_lam, const double* d_r, double mu, int V, double* d_w, int32_t* h_status) {
    hipStream_t st = h->stream;
    const int ne = n + (n & 1), ld = ne;
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
    const int np = ne / 2, rounds = ne - 1;
    const int jb = (np * np + TPB - 1) / TPB;
    if (ws.n != n || ws.batch != batch) {
        ws.release();
        ws.n = n;
        ws.batch = batch;
        LCHK(hipMalloc((void**)&ws.Bw, mb)); LCHK(hipMalloc((void**)&ws.W, mb)); LCHK(hipMalloc((void**)&ws.T1, mb));
        LCHK(hipMalloc((void**)&ws.C0, mb)); LCHK(hipMalloc((void**)&ws.C1, mb));
        LCHK(hipMalloc((void**)&ws.V0, mb)); LCHK(hipMalloc((void**)&ws.V1, mb));
        LCHK(hipMalloc((void**)&ws.dinv, sizeof(double) * vs * batch));
        LCHK(hipMalloc((void**)&ws.acc, sizeof(double) * 3 * batch));         // [sweep a | sweep b | ||C||_F^2]
        LCHK(hipMalloc((void**)&ws.coef, sizeof(double) * vs * batch));
        LCHK(hipMalloc((void**)&ws.flag, sizeof(int) * batch));
        LCHK(hipMalloc((void**)&ws.order, sizeof(int) * vs * batch));
        // two sweeps as one graph: 2 (ne - 1) rounds bring the ping-pong buffers back to where they started
        if (n > 1) {
            LCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            LCHK(hipMemsetAsync(ws.acc, 0, sizeof(double) * 2 * batch, st));
            double *Cc = ws.C0, *Cn = ws.C1, *Vc = ws.V0, *Vn = ws.V1;
            for (int sw = 0; sw < 2; ++sw)
                for (int r = 0; r < rounds; ++r) {
                    hipLaunchKernelGGL(jacobi_round_kernel, dim3(jb, 1, batch), dim3(TPB), 0, st, n, ne, ld, r, Cc, Cn, Vc, Vn,
                                       ws.acc + (size_t)sw * batch, ms);
                    double* t = Cc; Cc = Cn; Cn = t;
                    t = Vc; Vc = Vn; Vn = t;
                }
            LCHK(hipStreamEndCapture(st, &ws.graph));
            LCHK(hipGraphInstantiate(&ws.exec, ws.graph, nullptr, nullptr, 0));
        }
    }
    LCHK(hipMemsetAsync(ws.Bw, 0, mb, st));
    LCHK(hipMemsetAsync(ws.C0, 0, mb, st));
    LCHK(hipMemsetAsync(ws.C1, 0, mb, st));
    LCHK(hipMemsetAsync(ws.V1, 0, mb, st));
    LCHK(hipMemsetAsync(ws.flag, 0, sizeof(int) * batch, st));
    LCHK(hipMemsetAsync(ws.acc, 0, sizeof(double) * 3 * batch, st));
    // working copies with leading dimension ld (ghost row/column of an odd order stay zero)
    for (int z = 0; z < batch; ++z) {
        LCHK(hipMemcpy2DAsync(ws.Bw + z * ms, sizeof(double) * ld, d_B + (size_t)z * n * n, sizeof(double) * n,
                              sizeof(double) * n, n, hipMemcpyDeviceToDevice, st));
        LCHK(hipMemcpy2DAsync(ws.C0 + z * ms, sizeof(double) * ld, d_A + (size_t)z * n * n, sizeof(double) * n,
                              sizeof(double) * n, n, hipMemcpyDeviceToDevice, st));      // C0 holds A for now
    }
    hipLaunchKernelGGL(set_identity_kernel, dim3(gx, ne, batch), dim3(TPB), 0, st, ne, ld, ws.W, ms);
    hipLaunchKernelGGL(add_diag_kernel, dim3(gx, 1, batch), dim3(TPB), 0, st, n, ld, ws.Bw, reg, ms);      // apvast.py:24
    for (int kk = 0; kk < n; ++kk) {
        const int rows = n - kk - 1;
        hipLaunchKernelGGL(chol_inv_step_kernel, dim3(gx, rows > 0 ? rows : 1, batch), dim3(TPB), 0, st, n, ld, kk, ws.Bw,
                           ws.W, ws.dinv, ws.flag, ms, vs);
    }
    hipLaunchKernelGGL(scale_rows_kernel, dim3(gx, n, batch), dim3(TPB), 0, st, n, ld, ws.W, ws.dinv, ms, vs);
    const dim3 gg((n + 15) / 16, (n + 15) / 16, batch);
    hipLaunchKernelGGL((gemm_kernel<false, false>), gg, dim3(256), 0, st, n, ld, ws.W, ws.C0, ws.T1, ms);     // T1 = W A
    hipLaunchKernelGGL((gemm_kernel<false, true>), gg, dim3(256), 0, st, n, ld, ws.T1, ws.W, ws.C0, ms);      // C = T1 W^T
    hipLaunchKernelGGL(symmetrise_kernel, dim3(gx, n, batch), dim3(TPB), 0, st, n, ld, ws.C0, ms);
    hipLaunchKernelGGL(set_identity_kernel, dim3(gx, ne, batch), dim3(TPB), 0, st, ne, ld, ws.V0, ms);
    hipLaunchKernelGGL(frob2_kernel, dim3(64, 1, batch), dim3(TPB), 0, st, n, ld, ws.C0, ws.acc + 2 * batch, ms);
    std::vector<int> hflag(batch, 0);
    std::vector<double> hacc(3 * batch, 0.0);
    LCHK(hipMemcpyAsync(hflag.data(), ws.flag, sizeof(int) * batch, hipMemcpyDeviceToHost, st));
    LCHK(hipMemcpyAsync(hacc.data(), ws.acc, sizeof(double) * 3 * batch, hipMemcpyDeviceToHost, st));
    LCHK(hipStreamSynchronize(st));
    bool any_bad = false;
    for (int z = 0; z < batch; ++z) {
        h_status[z] = hflag[z] ? 1 : 0;
        any_bad = any_bad || hflag[z];
    }
    std::vector<double> norm2(hacc.begin() + 2 * batch, hacc.end());
    const int max_pairs = (h->cfg.max_sweeps > 0 ? h->cfg.max_sweeps : 30) / 2 + 1;
    bool converged = (n == 1);
    for (int it = 0; it < max_pairs && !converged; ++it) {
        LCHK(hipGraphLaunch(ws.exec, st));
        LCHK(hipMemcpyAsync(hacc.data(), ws.acc, sizeof(double) * 2 * batch, hipMemcpyDeviceToHost, st));
        LCHK(hipStreamSynchronize(st));
        converged = true;                     // judged on the second sweep of the pair
        for (int z = 0; z < batch; ++z)
            if (!hflag[z] && !(hacc[batch + z] <= 1e-20 * norm2[z])) converged = false;
    }
    if (!converged)
        for (int z = 0; z < batch; ++z)
            if (!hflag[z]) h_status[z] = 2;
    // after an even number of sweeps the current matrices are back in C0 / V0
    hipLaunchKernelGGL(rank_kernel, dim3(gx, 1, batch), dim3(TPB), 0, st, n, ld, ws.C0, d_lam, ws.order, ms, vs);
    hipLaunchKernelGGL((gemm_kernel<true, false>), gg, dim3(256), 0, st, n, ld, ws.W, ws.V0, ws.T1, ms);       // X = W^T Q
    hipLaunchKernelGGL(gather_cols_kernel, dim3(gx, n, batch), dim3(TPB), 0, st, n, ld, ws.T1, ws.order, d_U, ms, vs,
                       (size_t)n * n);
    if (d_r != nullptr && d_w != nullptr && V > 0) {
        hipLaunchKernelGGL(coef_kernel, dim3(gx, 1, batch), dim3(TPB), 0, st, n, d_U, d_lam, d_r, mu, ws.coef, (size_t)n * n, vs);
        hipLaunchKernelGGL(vast_prefix_kernel, dim3(gx, 1, batch), dim3(TPB), 0, st, n, V, d_U, ws.coef, d_w, (size_t)n * n, vs,
                           (size_t)V * n);
    }
    LCHK(hipStreamSynchronize(st));
    LCHK(hipGetLastError());
    if (any_bad) return apv_fail(h, APV_ERR_NOT_PD, "Matrix is not positive definite");
#undef LCHK
    return APV_OK;
}
