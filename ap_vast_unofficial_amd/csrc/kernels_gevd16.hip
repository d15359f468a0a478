// Order-16 specialisation of the fused subband update: one wavefront = one frequency bin.
//
//   stage 0  R_B, R_D = X^H X on the matrix cores (v_mfma_{f64,f32}_16x16x4): the [M][16] c64 slab of a
//            bin is read once, fully coalesced (lane l reads element 64 s + l), and the SAME register feeds
//            the A (X^H) and B (X) operands.                             apvast.py:329-364 per bin
//   stage 1  Cholesky of R_D + reg I, in LDS                             apvast.py:22-27
//   stage 2  C = L^-1 R_B L^-H                                           apvast.py:28-29
//   stage 3  cyclic Jacobi, round-robin order: the 64 lanes are the 8x8 grid of 2x2 blocks of C; lane
//            (a,b) holds rows {p_a,q_a} x columns {p_b,q_b} in registers for the round and the same
//            column pair of two fixed rows of V.                          apvast.py:30
//   stage 4-6 sort, X = L^-H Q, w_V = sum_{i<V} (x_i^H r)/(lam_i+mu) x_i  apvast.py:31-35, 406-414
//
// Everything between the load of X and the store of w stays in LDS/registers (13 KiB of LDS per wave).
#include "apv_internal.h"

#include <cstdlib>

#include "gevd16_common.h"

namespace {

template <typename T, bool FUSED, int JAC>
__global__ void __launch_bounds__(64) gevd16_kernel(const GevdParams p) {
    // zone program of a two-zone launch (blockIdx.y); the argument block itself stays in scalar registers
    const bool z1 = (blockIdx.y == 1);
    const float2* const pXB = z1 ? p.XB1 : p.XB;
    const float2* const pXD = z1 ? p.XD1 : p.XD;
    const float2* const pd = z1 ? p.d1 : p.d;
    void* const pw = z1 ? p.w1 : p.w;
    void* const plam = z1 ? p.lam1 : p.lam;
    int32_t* const pstatus = z1 ? p.status1 : p.status;
    using C = Cx<T>;
    __shared__ C sA[N * LD];
    __shared__ C sB[N * LD];
    __shared__ C sVstore[JAC == 1 ? 1 : N * LD];
    C* sV = (JAC == 1) ? sA : sVstore;     // register-resident Jacobi: C is dead by the time V is written
    __shared__ C sr[N];
    __shared__ C scoef[N];
    __shared__ T sDinv[N];
    __shared__ T sLam[N];
    __shared__ int sOrder[N];

    const int lane = threadIdx.x;
    const int k = blockIdx.x;
    int status = 0;

    // ---------------- stage 0 ----------------
    if constexpr (FUSED) {
        const size_t slab = (size_t)k * p.M * N;
        correlate16<T>(pXB + slab, pd + (size_t)k * p.M, p.M, sA, sr, lane);
        correlate16<T>(pXD + slab, nullptr, p.M, sB, sr, lane);
    } else {
        const C* RB = reinterpret_cast<const C*>(p.RB) + (size_t)k * N * N;
        const C* RD = reinterpret_cast<const C*>(p.RD) + (size_t)k * N * N;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int idx = lane + 64 * t, i = idx >> 4, j = idx & 15;
            sA[i * LD + j] = RB[idx];
            sB[i * LD + j] = RD[idx];
        }
        if (lane < N) sr[lane] = p.r ? reinterpret_cast<const C*>(p.r)[(size_t)k * N + lane] : mk<T>(0, 0);
    }
    wsync();
    if (p.debug_stop == 1) return;

    // ---------------- stage 1: dark loading + Cholesky (lower, in place) ----------------
    if (lane < N) {
        const C b = sB[lane * LD + lane];
        sB[lane * LD + lane] = mk<T>(b.x + (T)p.reg_dark, 0);
        sA[lane * LD + lane].y = 0;
    }
    wsync();
    {
        // lane (i = lane>>2, jq = lane&3) owns B[i][jq + 4t], t = 0..3
        const int i = lane >> 2, jq = lane & 3;
        for (int kk = 0; kk < N; ++kk) {
            const T dkk = sB[kk * LD + kk].x;
            if (!(dkk > (T)0) || !(dkk < (T)3.0e38)) { status = 1; break; }
            const T inv = rsq_full(dkk);
            if (lane == 0) sDinv[kk] = inv;
            const C lik = sB[i * LD + kk];                    // unscaled column entries
            C ljk[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) ljk[t] = sB[(jq + 4 * t) * LD + kk];
            wsync();
            const T inv2 = inv * inv;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = jq + 4 * t;
                if (j > kk && i >= j) {
                    // B[i][j] -= (B[i][kk]/d) * conj(B[j][kk]/d) * ... with d = sqrt(dkk): lik*conj(ljk)/dkk
                    C v = sB[i * LD + j];
                    v.x -= (lik.x * ljk[t].x + lik.y * ljk[t].y) * inv2;
                    v.y -= (lik.y * ljk[t].x - lik.x * ljk[t].y) * inv2;
                    sB[i * LD + j] = v;
                }
            }
            if (jq == 0 && i > kk) sB[i * LD + kk] = mk<T>(lik.x * inv, lik.y * inv);
            wsync();
        }
    }

    if (p.debug_stop == 2) return;
    if (status == 0) {
        // ---------------- stage 2: C = L^-1 A L^-H (two forward substitutions) ----------------
        // lane (i = lane>>2, jq) owns A[i][jq + 4t]
        const int i = lane >> 2, jq = lane & 3;
        for (int pass = 0; pass < 2; ++pass) {
            C a[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = sA[i * LD + jq + 4 * t];
            for (int kk = 0; kk < N; ++kk) {
                // row kk is final once scaled; rows below subtract L[i][kk] * row kk
                const T inv = sDinv[kk];
                if (i == kk) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        a[t] = mk<T>(a[t].x * inv, a[t].y * inv);
                        sA[kk * LD + jq + 4 * t] = a[t];
                    }
                }
                wsync();
                if (i > kk) {
                    const C l = sB[i * LD + kk];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const C y = sA[kk * LD + jq + 4 * t];
                        a[t].x -= l.x * y.x - l.y * y.y;
                        a[t].y -= l.x * y.y + l.y * y.x;
                    }
                }
            }
            wsync();
            if (pass == 0) {
                // rows are now final in sA; conjugate-transpose in place through registers
#pragma unroll
                for (int t = 0; t < 4; ++t) a[t] = sA[(jq + 4 * t) * LD + i];
                wsync();
#pragma unroll
                for (int t = 0; t < 4; ++t) sA[i * LD + jq + 4 * t] = mk<T>(a[t].x, -a[t].y);
                wsync();
            }
        }
        // symmetrise: C[i][j] = (C[i][j] + conj(C[j][i])) / 2
        {
            C u[4], l[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                u[t] = sA[i * LD + jq + 4 * t];
                l[t] = sA[(jq + 4 * t) * LD + i];
            }
            wsync();
#pragma unroll
            for (int t = 0; t < 4; ++t)
                sA[i * LD + jq + 4 * t] = mk<T>((T)0.5 * (u[t].x + l[t].x), (jq + 4 * t == i) ? (T)0 : (T)0.5 * (u[t].y - l[t].y));
        }
        if constexpr (JAC == 0) {       // V = I
#pragma unroll
            for (int t = 0; t < 4; ++t) sV[i * LD + jq + 4 * t] = mk<T>((jq + 4 * t == i) ? (T)1 : (T)0, (T)0);
        }
        wsync();

        if (p.debug_stop == 3) return;
        // ---------------- stage 3: cyclic Jacobi ----------------
        T nrm = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const C v = sA[i * LD + jq + 4 * t];
            nrm += v.x * v.x + v.y * v.y;
        }
        const T normF2 = wave_sum(nrm);

        const int a = lane >> 3, b = lane & 7;
        const int max_sweeps = p.max_sweeps > 0 ? p.max_sweeps : Prec<T>::max_sweeps;
        const T tol2 = p.sweep_tol2 > 0.0 ? (T)p.sweep_tol2 : Prec<T>::sweep_tol2;
        bool converged = false;
        if constexpr (JAC == 0) {
        const unsigned long long seqPa = c_seq.p[a], seqQa = c_seq.q[a];
        const unsigned long long seqPb = c_seq.p[b], seqQb = c_seq.q[b];
        const int v0 = (2 * a) * LD, v1 = (2 * a + 1) * LD;
        for (int sweep = 0; sweep < max_sweeps && !converged; ++sweep) {
            T off = 0;
            for (int r = 0; r < 15; ++r) {
                const int sh = 4 * r;
                const int pa = (int)(seqPa >> sh) & 15, qa = (int)(seqQa >> sh) & 15;
                const int pb = (int)(seqPb >> sh) & 15, qb = (int)(seqQb >> sh) & 15;
                // this lane's 2x2 block of C and 2x2 block of V
                const C xpp = sA[pa * LD + pb], xpq = sA[pa * LD + qb];
                const C xqp = sA[qa * LD + pb], xqq = sA[qa * LD + qb];
                const C v0p = sV[v0 + pb], v0q = sV[v0 + qb];
                const C v1p = sV[v1 + pb], v1q = sV[v1 + qb];
                // rotation of pair a from its diagonal block (8 lanes of a row compute the same thing)
                const T alpha = sA[pa * LD + pa].x, gamma = sA[qa * LD + qa].x;
                const C beta = sA[pa * LD + qa];
                const T b2 = beta.x * beta.x + beta.y * beta.y;
                off += b2;
                T ca = 1, sax = 0, say = 0;
                if (b2 > Prec<T>::tiny && b2 > Prec<T>::skip_rel * (alpha * alpha + gamma * gamma)) {
                    const T iab = rsq_full(b2);                       // 1/|beta|
                    const T tau = (gamma - alpha) * (T)0.5 * iab;
                    const T rrho = rsq_full((T)1 + tau * tau);        // 1/sqrt(1+tau^2)
                    const T c2 = (T)0.5 + (T)0.5 * fabs(tau) * rrho;  // cos^2(theta), in [1/2, 1]
                    const T r3 = rsq_full(c2);
                    ca = c2 * r3;
                    const T sr_ = copysign((T)0.5 * rrho * r3, tau);  // sin(theta), sign of tau
                    const T sc = sr_ * iab;
                    sax = beta.x * sc;
                    say = beta.y * sc;
                }
                // rotation of pair b lives in the lanes of row b
                const int src = 8 * b;
                const T cb = __shfl(ca, src, 64), sbx = __shfl(sax, src, 64), sby = __shfl(say, src, 64);
                wsync();        // every lane has read its operands before anyone overwrites them

                // columns: [x_p, x_q] J_b     (J = [[c, s], [-conj(s), c]])
                C ypp, ypq, yqp, yqq;
                ypp.x = cb * xpp.x - (sbx * xpq.x + sby * xpq.y);
                ypp.y = cb * xpp.y - (sbx * xpq.y - sby * xpq.x);
                ypq.x = cb * xpq.x + (sbx * xpp.x - sby * xpp.y);
                ypq.y = cb * xpq.y + (sbx * xpp.y + sby * xpp.x);
                yqp.x = cb * xqp.x - (sbx * xqq.x + sby * xqq.y);
                yqp.y = cb * xqp.y - (sbx * xqq.y - sby * xqq.x);
                yqq.x = cb * xqq.x + (sbx * xqp.x - sby * xqp.y);
                yqq.y = cb * xqq.y + (sbx * xqp.y + sby * xqp.x);
                // rows: J_a^H [y_p; y_q]      (J^H = [[c, -s], [conj(s), c]])
                C zpp, zpq, zqp, zqq;
                zpp.x = ca * ypp.x - (sax * yqp.x - say * yqp.y);
                zpp.y = ca * ypp.y - (sax * yqp.y + say * yqp.x);
                zpq.x = ca * ypq.x - (sax * yqq.x - say * yqq.y);
                zpq.y = ca * ypq.y - (sax * yqq.y + say * yqq.x);
                zqp.x = ca * yqp.x + (sax * ypp.x + say * ypp.y);
                zqp.y = ca * yqp.y + (sax * ypp.y - say * ypp.x);
                zqq.x = ca * yqq.x + (sax * ypq.x + say * ypq.y);
                zqq.y = ca * yqq.y + (sax * ypq.y - say * ypq.x);
                if (a == b) {
                    zpq = mk<T>(0, 0);
                    zqp = mk<T>(0, 0);
                    zpp.y = 0;
                    zqq.y = 0;
                }
                sA[pa * LD + pb] = zpp;
                sA[pa * LD + qb] = zpq;
                sA[qa * LD + pb] = zqp;
                sA[qa * LD + qb] = zqq;
                // V <- V J_b on rows 2a, 2a+1
                C w0p, w0q, w1p, w1q;
                w0p.x = cb * v0p.x - (sbx * v0q.x + sby * v0q.y);
                w0p.y = cb * v0p.y - (sbx * v0q.y - sby * v0q.x);
                w0q.x = cb * v0q.x + (sbx * v0p.x - sby * v0p.y);
                w0q.y = cb * v0q.y + (sbx * v0p.y + sby * v0p.x);
                w1p.x = cb * v1p.x - (sbx * v1q.x + sby * v1q.y);
                w1p.y = cb * v1p.y - (sbx * v1q.y - sby * v1q.x);
                w1q.x = cb * v1q.x + (sbx * v1p.x - sby * v1p.y);
                w1q.y = cb * v1q.y + (sbx * v1p.y + sby * v1p.x);
                sV[v0 + pb] = w0p;
                sV[v0 + qb] = w0q;
                sV[v1 + pb] = w1p;
                sV[v1 + qb] = w1q;
                wsync();
            }
            // every row of 8 lanes accumulated the same |beta_a|^2: the wave sum counts each pair 8 times
            const T tot = wave_sum(off) * (T)0.125;
            if (tot <= tol2 * normF2) converged = true;
        }
        if (lane < N) sLam[lane] = sA[lane * LD + lane].x;
        } else {
        // ---- register-resident variant: lane (a,b) keeps the 2x2 block {top_a,bot_a} x {top_b,bot_b} of C and the
        // columns {top_b,bot_b} of rows 2a, 2a+1 of V in registers for the whole iteration; data changes slot
        // through cross-lane XOR shuffles only.  Start layout: top_s = s, bot_s = 8 + s.
        // scale to ||C||_F in [1, 2): exact (power of two), undone on the eigenvalues
        const int sexp = (normF2 > (T)0) ? -(ilogb((double)normF2) / 2) : 0;
        const T scl = (T)ldexp(1.0, sexp), iscl = (T)ldexp(1.0, -sexp);
        C tt = sA[a * LD + b], tb = sA[a * LD + 8 + b], bt = sA[(8 + a) * LD + b], bb = sA[(8 + a) * LD + 8 + b];
        tt = mk<T>(tt.x * scl, tt.y * scl); tb = mk<T>(tb.x * scl, tb.y * scl);
        bt = mk<T>(bt.x * scl, bt.y * scl); bb = mk<T>(bb.x * scl, bb.y * scl);
        const T normS2 = normF2 * scl * scl;
        C v0t = mk<T>((2 * a == b) ? (T)1 : (T)0, 0), v0b = mk<T>((2 * a == 8 + b) ? (T)1 : (T)0, 0);
        C v1t = mk<T>((2 * a + 1 == b) ? (T)1 : (T)0, 0), v1b = mk<T>((2 * a + 1 == 8 + b) ? (T)1 : (T)0, 0);
        const bool diag = (a == b);
        int sweeps_done = 0;
        for (int sweep = 0; sweep < max_sweeps && !converged; ++sweep) {
            T off = 0;
            const XStep* sched = c_xsched[sweep & 1];
            for (int r = 0; r < 15; ++r) {
                const int tbit = sched[r].tbit, delta = sched[r].delta;
                if (tbit >= 0) {
                    const bool cb_ = (b >> tbit) & 1, ab_ = (a >> tbit) & 1;
                    const int pc = lane ^ (1 << tbit), pr = lane ^ (8 << tbit);
                    xchg(tt, tb, cb_, pc);      // columns
                    xchg(bt, bb, cb_, pc);
                    xchg(v0t, v0b, cb_, pc);
                    xchg(v1t, v1b, cb_, pc);
                    xchg(tt, bt, ab_, pr);      // rows
                    xchg(tb, bb, ab_, pr);
                }
                switch (delta) {
                    case 1: move_bottoms<1>(tb, bt, bb, v0b, v1b, lane); break;
                    case 2: move_bottoms<2>(tb, bt, bb, v0b, v1b, lane); break;
                    case 4: move_bottoms<4>(tb, bt, bb, v0b, v1b, lane); break;
                    default: break;
                }
                // rotation of this slot's pair, meaningful on the diagonal lanes
                if (diag) off += tb.x * tb.x + tb.y * tb.y;
                T c, sx, sy;
                rotation<T>(tt.x, bb.x, tb.x, tb.y, c, sx, sy);
                const int da = 9 * a, db = 9 * b;
                const T ca = __shfl(c, da, 64), sax = __shfl(sx, da, 64), say = __shfl(sy, da, 64);
                const T cb = __shfl(c, db, 64), sbx = __shfl(sx, db, 64), sby = __shfl(sy, db, 64);
                // columns: [x_t, x_b] J_b
                C ypp, ypq, yqp, yqq;
                ypp.x = cb * tt.x - (sbx * tb.x + sby * tb.y);
                ypp.y = cb * tt.y - (sbx * tb.y - sby * tb.x);
                ypq.x = cb * tb.x + (sbx * tt.x - sby * tt.y);
                ypq.y = cb * tb.y + (sbx * tt.y + sby * tt.x);
                yqp.x = cb * bt.x - (sbx * bb.x + sby * bb.y);
                yqp.y = cb * bt.y - (sbx * bb.y - sby * bb.x);
                yqq.x = cb * bb.x + (sbx * bt.x - sby * bt.y);
                yqq.y = cb * bb.y + (sbx * bt.y + sby * bt.x);
                // rows: J_a^H [y_t; y_b]
                tt.x = ca * ypp.x - (sax * yqp.x - say * yqp.y);
                tt.y = ca * ypp.y - (sax * yqp.y + say * yqp.x);
                tb.x = ca * ypq.x - (sax * yqq.x - say * yqq.y);
                tb.y = ca * ypq.y - (sax * yqq.y + say * yqq.x);
                bt.x = ca * yqp.x + (sax * ypp.x + say * ypp.y);
                bt.y = ca * yqp.y + (sax * ypp.y - say * ypp.x);
                bb.x = ca * yqq.x + (sax * ypq.x + say * ypq.y);
                bb.y = ca * yqq.y + (sax * ypq.y - say * ypq.x);
                if (diag) {         // the angle is only float-accurate: the residual beta' ~ 1e-7 beta is real data, keep it
                    tt.y = 0;
                    bb.y = 0;
                }
                // V <- V J_b
                C w0p, w0q, w1p, w1q;
                w0p.x = cb * v0t.x - (sbx * v0b.x + sby * v0b.y);
                w0p.y = cb * v0t.y - (sbx * v0b.y - sby * v0b.x);
                w0q.x = cb * v0b.x + (sbx * v0t.x - sby * v0t.y);
                w0q.y = cb * v0b.y + (sbx * v0t.y + sby * v0t.x);
                w1p.x = cb * v1t.x - (sbx * v1b.x + sby * v1b.y);
                w1p.y = cb * v1t.y - (sbx * v1b.y - sby * v1b.x);
                w1q.x = cb * v1b.x + (sbx * v1t.x - sby * v1t.y);
                w1q.y = cb * v1b.y + (sbx * v1t.y + sby * v1t.x);
                v0t = w0p; v0b = w0q; v1t = w1p; v1b = w1q;
            }
            ++sweeps_done;
            const T tot = wave_sum(off);
            if (tot <= tol2 * normS2) converged = true;
        }
        // after an odd number of sweeps slot s holds (2s, 2s+1), after an even number (s, 8+s)
        const bool nat = sweeps_done & 1;
        const int it_b = nat ? 2 * b : b, ib_b = nat ? 2 * b + 1 : 8 + b;
        wsync();
        sV[(2 * a) * LD + it_b] = v0t;
        sV[(2 * a) * LD + ib_b] = v0b;
        sV[(2 * a + 1) * LD + it_b] = v1t;
        sV[(2 * a + 1) * LD + ib_b] = v1b;
        if (diag) {
            sLam[it_b] = tt.x * iscl;
            sLam[ib_b] = bb.x * iscl;
        }
        }
        if (!converged) status = 2;

        // ---------------- stage 4: eigenvalues, descending order ----------------
        wsync();
        if (lane < N) {
            const T li = sLam[lane];
            int rank = 0;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const T lj = sLam[j];
                rank += (lj > li) || (lj == li && j < lane);
            }
            sOrder[rank] = lane;
        }

        // ---------------- stage 5: X = L^-H Q (backward substitution, rows in registers) ----------------
        {
            C x[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) x[t] = sV[i * LD + jq + 4 * t];
            for (int kk = N - 1; kk >= 0; --kk) {
                const T inv = sDinv[kk];
                if (i == kk) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        x[t] = mk<T>(x[t].x * inv, x[t].y * inv);
                        sV[kk * LD + jq + 4 * t] = x[t];
                    }
                }
                wsync();
                if (i < kk) {
                    const C l = sB[kk * LD + i];                     // conj(L[kk][i])
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const C y = sV[kk * LD + jq + 4 * t];
                        x[t].x -= l.x * y.x + l.y * y.y;
                        x[t].y -= l.x * y.y - l.y * y.x;
                    }
                }
            }
            wsync();
        }

        // ---------------- stage 6: coefficients (x_i^H r) / (lam_i + mu) ----------------
        if (lane < N) {
            T sx = 0, sy = 0;
#pragma unroll
            for (int l = 0; l < N; ++l) {
                const C v = sV[l * LD + lane], rr = sr[l];
                sx += v.x * rr.x + v.y * rr.y;
                sy += v.x * rr.y - v.y * rr.x;
            }
            const T den = (T)1 / (sLam[lane] + (T)p.mu);
            scoef[lane] = mk<T>(sx * den, sy * den);
        }
        wsync();
    }

    // ---------------- outputs ----------------
    if (lane < N) {
        T ax = 0, ay = 0;
        int done = 0;
        for (int t = 0; t < p.nV; ++t) {
            const int V = p.ranks[t];
            if (status != 1) {
                for (; done < V; ++done) {
                    const int c = sOrder[done];
                    const C cf = scoef[c], v = sV[lane * LD + c];
                    ax += cf.x * v.x - cf.y * v.y;
                    ay += cf.x * v.y + cf.y * v.x;
                }
            }
            const size_t o = ((size_t)k * p.nV + t) * N + lane;
            if (p.out_c128) reinterpret_cast<double2*>(pw)[o] = make_double2((double)ax, (double)ay);
            else reinterpret_cast<float2*>(pw)[o] = make_float2((float)ax, (float)ay);
        }
        if (plam != nullptr) {
            const T lv = (status != 1) ? sLam[sOrder[lane]] : (T)0;
            if (p.out_c128) reinterpret_cast<double*>(plam)[(size_t)k * N + lane] = (double)lv;
            else reinterpret_cast<float*>(plam)[(size_t)k * N + lane] = (float)lv;
        }
    }
    if (p.U != nullptr) {
        C* U = reinterpret_cast<C*>(p.U) + (size_t)k * N * N;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int idx = lane + 64 * t, i = idx >> 4, j = idx & 15;
            U[idx] = (status != 1) ? sV[i * LD + sOrder[j]] : mk<T>(0, 0);
        }
    }
    if (pstatus != nullptr && lane == 0) pstatus[k] = status;
}

}  // namespace

// n == 16, absolute dark loading, no bright loading: the fast path.  Returns hipErrorNotSupported otherwise.
hipError_t apv_launch_gevd16(const GevdParams& p, int compute_dtype, bool fused, hipStream_t s) {
    if (p.n != 16 || p.reg_mode != APV_REG_ABS || p.reg_bright != 0.0) return hipErrorNotSupported;
    if (p.K <= 0) return hipSuccess;
    static const int variant = [] {
        const char* e = getenv("APV_GEVD16_JACOBI");       // "lds" | "reg" (A/B switch for profiling)
        return (e && e[0] == 'l') ? 0 : 1;
    }();
#define APV_L16(T, F, J) hipLaunchKernelGGL((gevd16_kernel<T, F, J>), dim3(p.K, p.n_zones > 1 ? 2 : 1), dim3(64), 0, s, p)
    if (compute_dtype == APV_F64) {
        if (variant == 0) { if (fused) APV_L16(double, true, 0); else APV_L16(double, false, 0); }
        else              { if (fused) APV_L16(double, true, 1); else APV_L16(double, false, 1); }
    } else {
        if (variant == 0) { if (fused) APV_L16(float, true, 0); else APV_L16(float, false, 0); }
        else              { if (fused) APV_L16(float, true, 1); else APV_L16(float, false, 1); }
    }
#undef APV_L16
    return hipGetLastError();
}
