// Order-16 fused subband update, one wavefront per frequency bin; every dense 16x16x16 contraction is on the
// matrix cores.
//
//   stage 0  R_B, R_D = X^H X, r = X_B^H d        MFMA 16x16x4 (f64 or f32), slab read once, coalesced
//   stage 1  Cholesky of R_D + reg I TOGETHER WITH W = L^-1 (elimination applied to [B | I]); rows of B and W
//            live in registers, one column / one row is broadcast through LDS per step      apvast.py:22-27
//   stage 2  C = W R_B W^H                         two complex MFMA products                 apvast.py:28-29
//   stage 3  cyclic Jacobi, register resident, XOR pairing schedule (gevd16_common.h)        apvast.py:30
//            float64: a float32 pre-solve on packed math, its eigenvector matrix re-orthonormalised to first order
//            and C re-formed on the f64 MFMA, then the double sweeps (one, at the default tolerance)
//   stage 4  sort                                                                            apvast.py:32-35
//   stage 5  X = W^H Q                             one complex MFMA product                  apvast.py:31
//   stage 6  w_V = sum_{i<V} (x_i^H r)/(lam_i+mu) x_i                                        apvast.py:406-414
//
// Against a textbook arrangement this removes 48 of the 64 sequential substitution steps (the two forward
// substitutions and the backward one become three MFMA products) and needs two LDS matrices (10 KiB per wave).
#include "apv_internal.h"

#include <cstdlib>

#include "gevd16_common.h"

namespace {

// complex 16x16x16 product on the matrix cores.  fa(i, k) / fb(k, j) fetch operand elements; out[t] is the
// element (mfma_row<T>(lane, t), lane & 15).
template <typename T> __device__ __forceinline__ int mfma_row(int lane, int t);
template <> __device__ __forceinline__ int mfma_row<double>(int lane, int t) { return (lane >> 4) + 4 * t; }
template <> __device__ __forceinline__ int mfma_row<float>(int lane, int t) { return 4 * (lane >> 4) + t; }

template <typename FA, typename FB>
__device__ __forceinline__ void cmm16(FA fa, FB fb, int lane, Cx<double> out[4]) {
    d4 re = {0, 0, 0, 0}, im = {0, 0, 0, 0};
    const int rc = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const Cx<double> a = fa(rc, 4 * s + kq), b = fb(4 * s + kq, rc);
        re = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, re, 0, 0, 0);
        re = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.y, b.y, re, 0, 0, 0);
        im = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.y, im, 0, 0, 0);
        im = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.x, im, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) out[t] = mk<double>(re[t], im[t]);
}
template <typename FA, typename FB>
__device__ __forceinline__ void cmm16(FA fa, FB fb, int lane, Cx<float> out[4]) {
    f4 re = {0, 0, 0, 0}, im = {0, 0, 0, 0};
    const int rc = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const Cx<float> a = fa(rc, 4 * s + kq), b = fb(4 * s + kq, rc);
        re = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, re, 0, 0, 0);
        re = __builtin_amdgcn_mfma_f32_16x16x4f32(-a.y, b.y, re, 0, 0, 0);
        im = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.y, im, 0, 0, 0);
        im = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.x, im, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) out[t] = mk<float>(re[t], im[t]);
}

// The register-resident cyclic Jacobi of stage 3 on the 2 x 2 blocks (tt, tb; bt, bb) of this lane and its two rows
// of V, in precision TT.  Runs sweeps until one of them meets sum |pivot|^2 <= tol2 normS2 (that sweep is the last) or
// max_sweeps is reached; returns the number of sweeps done (its parity says which slot layout the blocks are left in).
template <typename TT>
__device__ __forceinline__ int jacobi16_sweeps(Cx<TT>& tt_, Cx<TT>& tb_, Cx<TT>& bt_, Cx<TT>& bb_, Cx<TT>& v0t_, Cx<TT>& v0b_,
                                               Cx<TT>& v1t_, Cx<TT>& v1b_, TT (*srot)[4], int lane, TT tol2, TT normS2,
                                               int max_sweeps, bool& converged_) {
    using CC = Cx<TT>;
    const int a = lane >> 3, b = lane & 7;
    const bool diag = (a == b);
    int sweeps_done = 0;
    bool converged = false;
    // work on local copies: the blocks must stay in registers (by-reference structs end up in scratch otherwise)
    CC tt = tt_, tb = tb_, bt = bt_, bb = bb_, v0t = v0t_, v0b = v0b_, v1t = v1t_, v1b = v1b_;
    for (int sweep = 0; sweep < max_sweeps && !converged; ++sweep) {
        TT off = 0;
        // the schedule as two nibble strings in scalar registers (a table in memory costs a load per round)
        const unsigned long long dseq = (sweep & 1) ? XS_DELTA1 : XS_DELTA0;
        const unsigned long long tseq = (sweep & 1) ? XS_TBIT1 : XS_TBIT0;
        for (int r = 0; r < 15; ++r) {
            const int delta = (int)((dseq >> (4 * r)) & 15), tbit = (int)((tseq >> (4 * r)) & 15) - 1;
            if (tbit >= 0) {
                // columns first, then rows; the row exchanges and the bit-2 column exchange are masked lane swaps
                if (tbit == 2) {
                    cxswap_col4(tt, tb);
                    cxswap_col4(bt, bb);
                    cxswap_col4(v0t, v0b);
                    cxswap_col4(v1t, v1b);
                    cxswap_row<2>(tt, bt);
                    cxswap_row<2>(tb, bb);
                } else {
                    const bool cb_ = (b >> tbit) & 1;
                    const int pc = lane ^ (1 << tbit);
                    xchg(tt, tb, cb_, pc);
                    xchg(bt, bb, cb_, pc);
                    xchg(v0t, v0b, cb_, pc);
                    xchg(v1t, v1b, cb_, pc);
                    if (tbit == 1) {
                        cxswap_row<1>(tt, bt);
                        cxswap_row<1>(tb, bb);
                    } else {
                        cxswap_row<0>(tt, bt);
                        cxswap_row<0>(tb, bb);
                    }
                }
            }
            switch (delta) {
                case 1: move_bottoms<1>(tb, bt, bb, v0b, v1b, lane); break;
                case 2: move_bottoms<2>(tb, bt, bb, v0b, v1b, lane); break;
                case 4: move_bottoms<4>(tb, bt, bb, v0b, v1b, lane); break;
                default: break;
            }
            if (diag) off += tb.x * tb.x + tb.y * tb.y;
            TT c, sx, sy;
            rotation<TT>(tt.x, bb.x, tb.x, tb.y, c, sx, sy);
            TT ca, sax, say, cb, sbx, sby;
            if constexpr (sizeof(TT) == 8) {
                // double: the eight rotations go through LDS (two wide reads per lane instead of twelve ds_bpermute)
                if (diag) {
                    srot[a][0] = c;
                    srot[a][1] = sx;
                    srot[a][2] = sy;
                }
                wsync();
                ca = srot[a][0]; sax = srot[a][1]; say = srot[a][2];
                cb = srot[b][0]; sbx = srot[b][1]; sby = srot[b][2];
            } else {
                const int da = 9 * a, db = 9 * b;
                ca = __shfl(c, da, 64); sax = __shfl(sx, da, 64); say = __shfl(sy, da, 64);
                cb = __shfl(c, db, 64); sbx = __shfl(sx, db, 64); sby = __shfl(sy, db, 64);
            }
            const CC sa = mk<TT>(sax, say), sb = mk<TT>(sbx, sby);
            CC ypp, ypq, yqp, yqq;
            rot_cols<TT>(cb, sb, tt, tb, ypp, ypq);
            rot_cols<TT>(cb, sb, bt, bb, yqp, yqq);
            rot_rows<TT>(ca, sa, ypp, yqp, tt, bt);
            rot_rows<TT>(ca, sa, ypq, yqq, tb, bb);
            if (diag) {         // the angle is float-accurate: the residual beta' ~ 1e-7 beta is real data, keep it
                tt.y = 0;
                bb.y = 0;
            }
            CC w0p, w0q, w1p, w1q;
            rot_cols<TT>(cb, sb, v0t, v0b, w0p, w0q);
            rot_cols<TT>(cb, sb, v1t, v1b, w1p, w1q);
            v0t = w0p; v0b = w0q; v1t = w1p; v1b = w1q;
        }
        ++sweeps_done;
        const TT tot = wave_sum(off);
        if (tot <= tol2 * normS2) converged = true;
    }
    tt_ = tt; tb_ = tb; bt_ = bt; bb_ = bb; v0t_ = v0t; v0b_ = v0b; v1t_ = v1t; v1b_ = v1b;
    converged_ = converged;
    return sweeps_done;
}

// XT: element type of the fused input slabs (float2 = c64, double2 = c128: the float64 streaming front-end)
template <typename T, bool FUSED, typename XT>
__device__ __forceinline__ void gevd16m_body(const GevdParams& p) {
    // zone program of a two-zone launch (blockIdx.y); the argument block itself stays in scalar registers
    const bool z1 = (blockIdx.y == 1);
    const XT* const pXB = reinterpret_cast<const XT*>(z1 ? p.XB1 : p.XB);
    const XT* const pXD = reinterpret_cast<const XT*>(z1 ? p.XD1 : p.XD);
    const XT* const pd = reinterpret_cast<const XT*>(z1 ? p.d1 : p.d);
    void* const pw = z1 ? p.w1 : p.w;
    void* const plam = z1 ? p.lam1 : p.lam;
    int32_t* const pstatus = z1 ? p.status1 : p.status;
    using C = Cx<T>;
    __shared__ C sA[N * LD];       // R_B -> W R_B -> C -> Q (eigenvectors of C) -> X
    __shared__ C sB[N * LD];       // R_D -> W = L^-1
    __shared__ C scol[2][N];       // Cholesky: current column of B (double buffered)
    __shared__ C swr[2][N];        // Cholesky: current row of W
    __shared__ C sr[N];
    __shared__ C scoef[N];
    // The Cholesky staging is dead after stage 1; the later stages' small arrays live in it, which keeps the
    // workgroup at 10 240 B of LDS = exactly 16 workgroups (4 waves per SIMD) per CU.
    T* const sLam = reinterpret_cast<T*>(&scol[0][0]);                 // [N]
    int* const sOrder = reinterpret_cast<int*>(&scol[1][0]);           // [N]
    T (*const srot)[4] = reinterpret_cast<T(*)[4]>(&swr[0][0]);        // Jacobi: (c, s.x, s.y) of the eight rotations of a round

    const int lane = threadIdx.x;
    const int k = blockIdx.x;
    int status = 0;

    // ---------------- stage 0 ----------------
    if constexpr (FUSED) {
        const size_t slab = (size_t)k * p.M * N;
        correlate16<T, XT>(pXB + slab, pd + (size_t)k * p.M, p.M, sA, sr, lane);
        correlate16<T, XT>(pXD + slab, (const XT*)nullptr, p.M, sB, sr, lane);
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

    // ---------------- stage 1: Cholesky of B + reg I with W = L^-1 ----------------
    // lane (i = lane>>2, jq = lane&3) owns B[i][jq+4t] and W[i][jq+4t], t = 0..3, in registers
    const int i = lane >> 2, jq = lane & 3;
    C brow[4], wrow[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = jq + 4 * t;
        brow[t] = sB[i * LD + j];
        if (j == i) brow[t] = mk<T>(brow[t].x + (T)p.reg_dark, 0);
        wrow[t] = mk<T>((j == i) ? (T)1 : (T)0, 0);
    }
#pragma unroll
    for (int kk = 0; kk < N; ++kk) {
        const int buf = kk & 1;
        if (jq == (kk & 3)) scol[buf][i] = brow[kk >> 2];                       // column kk of the working B
        if (i == kk) {
#pragma unroll
            for (int t = 0; t < 4; ++t) swr[buf][jq + 4 * t] = wrow[t];         // row kk of the working W
        }
        wsync();
        const T dkk = scol[buf][kk].x;
        if (!(dkk > (T)0) || !(dkk < (T)3.0e38)) { status = 1; break; }         // uniform: same LDS word for all lanes
        const T inv = rsq_full(dkk), inv2 = inv * inv;
        const C li = scol[buf][i];
        if (i > kk) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = jq + 4 * t;
                if (j > kk && j <= i) {                  // B[i][j] -= B[i][kk] conj(B[j][kk]) / d
                    const C lj = scol[buf][j];
                    brow[t].x -= (li.x * lj.x + li.y * lj.y) * inv2;
                    brow[t].y -= (li.y * lj.x - li.x * lj.y) * inv2;
                }
                if (j <= kk) {                           // W[i][j] -= (B[i][kk]/sqrt d) (W[kk][j]/sqrt d)
                    const C wk = swr[buf][j];
                    wrow[t].x -= (li.x * wk.x - li.y * wk.y) * inv2;
                    wrow[t].y -= (li.x * wk.y + li.y * wk.x) * inv2;
                }
            }
        } else if (i == kk) {
#pragma unroll
            for (int t = 0; t < 4; ++t) wrow[t] = mk<T>(wrow[t].x * inv, wrow[t].y * inv);
        }
    }
    wsync();
    if (p.debug_stop == 2) return;

    if (status == 0) {
        // W -> sB (R_D is spent)
#pragma unroll
        for (int t = 0; t < 4; ++t) sB[i * LD + jq + 4 * t] = wrow[t];
        wsync();
        // ---------------- stage 2: C = W A W^H ----------------
        C acc[4];
        cmm16([&](int r, int kx) { return sB[r * LD + kx]; }, [&](int kx, int c) { return sA[kx * LD + c]; }, lane, acc);
        wsync();
        const int col = lane & 15;
#pragma unroll
        for (int t = 0; t < 4; ++t) sA[mfma_row<T>(lane, t) * LD + col] = acc[t];          // T1 = W A
        wsync();
        cmm16([&](int r, int kx) { return sA[r * LD + kx]; },
              [&](int kx, int c) { const C w = sB[c * LD + kx]; return mk<T>(w.x, -w.y); }, lane, acc);
        wsync();
        T nrm = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = mfma_row<T>(lane, t);
            if (row == col) acc[t].y = 0;
            sA[row * LD + col] = acc[t];                                                    // C
            nrm += acc[t].x * acc[t].x + acc[t].y * acc[t].y;
        }
        const T normF2 = wave_sum(nrm);
        wsync();
        if (p.debug_stop == 3) return;

        // ---------------- stage 3: register-resident Jacobi, XOR schedule ----------------
        const int a = lane >> 3, b = lane & 7;
        const int max_sweeps = p.max_sweeps > 0 ? p.max_sweeps : Prec<T>::max_sweeps;
        const T tol2 = p.sweep_tol2 > 0.0 ? (T)p.sweep_tol2 : Prec<T>::sweep_tol2;
        bool converged = false;
        const int sexp = (normF2 > (T)0) ? -(ilogb((double)normF2) / 2) : 0;
        const T scl = (T)ldexp(1.0, sexp), iscl = (T)ldexp(1.0, -sexp);
        const T normS2 = normF2 * scl * scl;
        bool v_in_lds = false;
        if constexpr (sizeof(T) == 8) {
            // ---- float32 pre-solve (debug_stop == 4 skips it: double sweeps only, for A/B timing) -------------------
            // The sweeps are the cost of the kernel and the packed-float ones are less than half as expensive, so C is
            // first diagonalised in float: V32 with V32^H C V32 diagonal to ~1e-7.  V32 is unitary only to 1e-7, so it is
            // not used as it is: with E = V32^H V32 - I,  V' = V32 (I - E/2) is unitary to E^2 ~ 1e-14 and
            // C' = V'^H C V' = C1 - (E C1 + C1 E)/2,  C1 = V32^H C V32  (five complex 16 x 16 x 16 MFMA products).  The
            // double sweeps then start from (C', V'): one is enough for the default tolerance.  W waits in registers.
            if (p.debug_stop != 4) {
                using CF = Cx<float>;
                // looser than the float kernel's own 1e-8: a double sweep follows anyway, so the float sweep that would only
                // confirm convergence is not run (1e-6 measured best; 1e-5 leaves more bins needing a second double sweep)
                constexpr float kPresolveTol2 = 1e-6f;
                auto ldf = [&](int r, int c) { const C v = sA[r * LD + c]; return mk<float>((float)(v.x * scl), (float)(v.y * scl)); };
                CF ftt = ldf(a, b), ftb = ldf(a, 8 + b), fbt = ldf(8 + a, b), fbb = ldf(8 + a, 8 + b);
                CF f0t = mk<float>((2 * a == b) ? 1.f : 0.f, 0.f), f0b = mk<float>((2 * a == 8 + b) ? 1.f : 0.f, 0.f);
                CF f1t = mk<float>((2 * a + 1 == b) ? 1.f : 0.f, 0.f), f1b = mk<float>((2 * a + 1 == 8 + b) ? 1.f : 0.f, 0.f);
                bool fconv = false;
                const int fs = jacobi16_sweeps<float>(ftt, ftb, fbt, fbb, f0t, f0b, f1t, f1b, (float (*)[4]) nullptr, lane,
                                                      kPresolveTol2, (float)normS2, Prec<float>::max_sweeps, fconv);
                const bool fnat = fs & 1;
                const int fit = fnat ? 2 * b : b, fib = fnat ? 2 * b + 1 : 8 + b;
                const int mcol = lane & 15;
                auto cj = [](C w) { return mk<T>(w.x, -w.y); };
                wsync();
                sB[(2 * a) * LD + fit] = mk<T>((T)f0t.x, (T)f0t.y);                 // V32 takes W's place
                sB[(2 * a) * LD + fib] = mk<T>((T)f0b.x, (T)f0b.y);
                sB[(2 * a + 1) * LD + fit] = mk<T>((T)f1t.x, (T)f1t.y);
                sB[(2 * a + 1) * LD + fib] = mk<T>((T)f1b.x, (T)f1b.y);
                wsync();
                C accT[4], accG[4], accC[4], accV[4], accE[4];
                cmm16([&](int r, int kx) { return sA[r * LD + kx]; }, [&](int kx, int c) { return sB[kx * LD + c]; }, lane, accT);       // C V
                cmm16([&](int r, int kx) { return cj(sB[kx * LD + r]); }, [&](int kx, int c) { return sB[kx * LD + c]; }, lane, accG);   // V^H V
                wsync();
#pragma unroll
                for (int t = 0; t < 4; ++t) sA[mfma_row<T>(lane, t) * LD + mcol] = accT[t];
                wsync();
                cmm16([&](int r, int kx) { return cj(sB[kx * LD + r]); }, [&](int kx, int c) { return sA[kx * LD + c]; }, lane, accC);   // C1 = V^H C V
                wsync();
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int row = mfma_row<T>(lane, t);
                    sA[row * LD + mcol] = mk<T>(accG[t].x - (row == mcol ? (T)1 : (T)0), accG[t].y);                                     // E
                }
                wsync();
                cmm16([&](int r, int kx) { return sB[r * LD + kx]; }, [&](int kx, int c) { return sA[kx * LD + c]; }, lane, accV);       // V E
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const C v = sB[mfma_row<T>(lane, t) * LD + mcol];
                    accV[t] = mk<T>(v.x - (T)0.5 * accV[t].x, v.y - (T)0.5 * accV[t].y);                                                 // V'
                }
                wsync();
#pragma unroll
                for (int t = 0; t < 4; ++t) sB[mfma_row<T>(lane, t) * LD + mcol] = accC[t];
                wsync();
                cmm16([&](int r, int kx) { return sA[r * LD + kx]; }, [&](int kx, int c) { return sB[kx * LD + c]; }, lane, accE);       // E C1
                wsync();
#pragma unroll
                for (int t = 0; t < 4; ++t) sA[mfma_row<T>(lane, t) * LD + mcol] = accE[t];
                wsync();
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int row = mfma_row<T>(lane, t);
                    const C h = cj(sA[mcol * LD + row]);                                                                                // (E C1)^H = C1 E
                    accC[t] = mk<T>(accC[t].x - (T)0.5 * (accE[t].x + h.x), accC[t].y - (T)0.5 * (accE[t].y + h.y));
                    if (row == mcol) accC[t].y = 0;
                }
                wsync();
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    sA[mfma_row<T>(lane, t) * LD + mcol] = accC[t];                                                                     // C'
                    sB[mfma_row<T>(lane, t) * LD + mcol] = accV[t];                                                                     // V'
                }
                wsync();
                v_in_lds = true;
            }
        }
        C tt = sA[a * LD + b], tb = sA[a * LD + 8 + b], bt = sA[(8 + a) * LD + b], bb = sA[(8 + a) * LD + 8 + b];
        tt = mk<T>(tt.x * scl, tt.y * scl); tb = mk<T>(tb.x * scl, tb.y * scl);
        bt = mk<T>(bt.x * scl, bt.y * scl); bb = mk<T>(bb.x * scl, bb.y * scl);
        C v0t = mk<T>((2 * a == b) ? (T)1 : (T)0, 0), v0b = mk<T>((2 * a == 8 + b) ? (T)1 : (T)0, 0);
        C v1t = mk<T>((2 * a + 1 == b) ? (T)1 : (T)0, 0), v1b = mk<T>((2 * a + 1 == 8 + b) ? (T)1 : (T)0, 0);
        if (v_in_lds) {
            v0t = sB[(2 * a) * LD + b]; v0b = sB[(2 * a) * LD + 8 + b];
            v1t = sB[(2 * a + 1) * LD + b]; v1b = sB[(2 * a + 1) * LD + 8 + b];
            wsync();
#pragma unroll
            for (int t = 0; t < 4; ++t) sB[i * LD + jq + 4 * t] = wrow[t];          // W back in place for stage 5
        }
        const bool diag = (a == b);
        const int sweeps_done = jacobi16_sweeps<T>(tt, tb, bt, bb, v0t, v0b, v1t, v1b, srot, lane, tol2, normS2, max_sweeps, converged);
        if (!converged) status = 2;
        // after an odd number of sweeps slot s holds (2s, 2s+1), after an even number (s, 8+s)
        const bool nat = sweeps_done & 1;
        const int it_b = nat ? 2 * b : b, ib_b = nat ? 2 * b + 1 : 8 + b;
        wsync();
        sA[(2 * a) * LD + it_b] = v0t;
        sA[(2 * a) * LD + ib_b] = v0b;
        sA[(2 * a + 1) * LD + it_b] = v1t;
        sA[(2 * a + 1) * LD + ib_b] = v1b;
        if (diag) {
            sLam[it_b] = tt.x * iscl;
            sLam[ib_b] = bb.x * iscl;
        }
        wsync();

        // ---------------- stage 4: descending order ----------------
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

        // ---------------- stage 5: X = W^H Q ----------------
        cmm16([&](int r, int kx) { const C w = sB[kx * LD + r]; return mk<T>(w.x, -w.y); },
              [&](int kx, int c) { return sA[kx * LD + c]; }, lane, acc);
        wsync();
#pragma unroll
        for (int t = 0; t < 4; ++t) sA[mfma_row<T>(lane, t) * LD + col] = acc[t];
        wsync();

        // ---------------- stage 6: coefficients (x_i^H r) / (lam_i + mu) ----------------
        if (lane < N) {
            T sx = 0, sy = 0;
#pragma unroll
            for (int l = 0; l < N; ++l) {
                const C v = sA[l * LD + lane], rr = sr[l];
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
                    const C cf = scoef[c], v = sA[lane * LD + c];
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
            const int idx = lane + 64 * t, ii = idx >> 4, j = idx & 15;
            U[idx] = (status != 1) ? sA[ii * LD + sOrder[j]] : mk<T>(0, 0);
        }
    }
    if (pstatus != nullptr && lane == 0) pstatus[k] = status;
}

// The double kernel is held to four waves per SIMD (its float32 pre-solve and the re-orthonormalisation products would
// otherwise raise the register count past 128 and cost a wave); the float kernel is left to the compiler.
template <typename T, bool FUSED, typename XT>
__global__ void __launch_bounds__(64) gevd16m_kernel(const GevdParams p) { gevd16m_body<T, FUSED, XT>(p); }
template <bool FUSED, typename XT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) gevd16m_kernel_f64(const GevdParams p) {
    gevd16m_body<double, FUSED, XT>(p);
}

}  // namespace

hipError_t apv_launch_gevd16m(const GevdParams& p, int compute_dtype, bool fused, hipStream_t s) {
    if (p.n != 16 || p.reg_mode != APV_REG_ABS || p.reg_bright != 0.0) return hipErrorNotSupported;
    if (p.K <= 0) return hipSuccess;
    const dim3 grid(p.K, p.n_zones > 1 ? 2 : 1);
    const bool xd = fused && p.x_c128;
    if (compute_dtype == APV_F64) {
        if (xd) hipLaunchKernelGGL((gevd16m_kernel_f64<true, double2>), grid, dim3(64), 0, s, p);
        else if (fused) hipLaunchKernelGGL((gevd16m_kernel_f64<true, float2>), grid, dim3(64), 0, s, p);
        else hipLaunchKernelGGL((gevd16m_kernel_f64<false, float2>), grid, dim3(64), 0, s, p);
    } else {
        if (xd) hipLaunchKernelGGL((gevd16m_kernel<float, true, double2>), grid, dim3(64), 0, s, p);
        else if (fused) hipLaunchKernelGGL((gevd16m_kernel<float, true, float2>), grid, dim3(64), 0, s, p);
        else hipLaunchKernelGGL((gevd16m_kernel<float, false, float2>), grid, dim3(64), 0, s, p);
    }
    return hipGetLastError();
}
