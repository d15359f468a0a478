// Order-16 fused subband update, one wavefront per frequency bin; every dense 16x16x16 contraction is on the
// matrix cores.
//
//   stage 0  R_B, R_D = X^H X, r = X_B^H d        MFMA 16x16x4 (f64 or f32), slab read once, coalesced
//   stage 1  Cholesky of R_D + reg I TOGETHER WITH W = L^-1 (elimination applied to [B | I]); rows of B and W
//            live in registers, one column / one row is broadcast through LDS per step      apvast.py:22-27
//   stage 2  C = W R_B W^H                         two complex MFMA products                 apvast.py:28-29
//   stage 3  eigenvectors of C                                                                apvast.py:30
//            float64: a float32 pre-solve on packed math -- ONE-SIDED Jacobi on the float Cholesky factor of C (column
//            rotations only, gevd16_common.h) -- then one or two refinement steps of its eigenvector matrix on the f64 MFMA
//            (four complex products each, Ogita & Aishima 2018); a wave whose spectrum has a gap too narrow for that
//            orthonormalises exactly, re-forms C on the MFMA and runs register-resident double sweeps (XOR pairing schedule)
//            float32: the one-sided Jacobi is the solve
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

// XT: element type of the fused input slabs (float2 = c64, double2 = c128: the float64 streaming front-end)
// DBG: the diagnostic instantiation.  It alone carries the run-time `debug_stop` tests (stage cuts and A/B switches of the
// probes under tools/probes/) and the in-kernel stage stamps (p.stamps); in the product instantiation `dstop` is the constant 0
// and all of it folds away, the round-2a two-sided pre-solve included.
// HOPS: the launch covers several hops of a chunk (blockIdx.z = hop): every operand moves on by its byte stride per hop (scalar
// arithmetic on the argument block; the instantiations without it are untouched)
template <typename T, bool FUSED, typename XT, bool DBG, int GS = 1, bool HOPS = false>
__device__ __forceinline__ void gevd16m_body(const GevdParams& p, const int k, const bool z1, const int hop = 0) {
    const int dstop = DBG ? p.debug_stop : 0;
    // stage stamps (diagnostic build only): s_memtime of lane 0 at the stage boundaries, 8 per bin, into a buffer of their own
    auto stamp = [&](int i) {
        if constexpr (DBG) {
            if (p.stamps != nullptr && threadIdx.x == 0)
                p.stamps[((size_t)(z1 ? 1 : 0) * p.K + k) * 16 + i] = __builtin_amdgcn_s_memtime();
        }
    };
    stamp(0);
    if constexpr (DBG) { if (p.stamps != nullptr && threadIdx.x == 0) p.stamps[((size_t)(z1 ? 1 : 0) * p.K + k) * 16 + 15] = __builtin_amdgcn_s_memrealtime(); }
    // zone program of a two-zone launch (z1); the argument block itself stays in scalar registers
    auto at_hop = [&](const void* base, size_t stride) -> const char* {
        const char* b = static_cast<const char*>(base);
        if constexpr (HOPS) return b ? b + (size_t)hop * stride : b;
        else return b;
    };
    const XT* const pXB = reinterpret_cast<const XT*>(at_hop(z1 ? p.XB1 : p.XB, p.hop_X));
    const XT* const pXD = reinterpret_cast<const XT*>(at_hop(z1 ? p.XD1 : p.XD, p.hop_X));
    const XT* const pd = reinterpret_cast<const XT*>(at_hop(z1 ? p.d1 : p.d, p.hop_d));
    void* const pw = const_cast<char*>(at_hop(z1 ? p.w1 : p.w, p.hop_w));
    void* const plam = const_cast<char*>(at_hop(z1 ? p.lam1 : p.lam, p.hop_lam));
    int32_t* const pstatus = reinterpret_cast<int32_t*>(const_cast<char*>(at_hop(z1 ? p.status1 : p.status, p.hop_status)));
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
    T* const sPiv = reinterpret_cast<T*>(&scoef[0]);                   // [N] stage 1: the pivots (scoef is idle until stage 6)

    const int lane = threadIdx.x;
    int status = 0;
    // One wave is a long dependent chain; in the per-hop streaming pipeline the launch shares the chip with the next hop's
    // transforms, whose waves would otherwise take every other issue slot: this wave goes first.  (No effect when the kernel runs
    // alone.)  The chunked whole-signal path asks for the opposite (yield_issue): there the transforms are the longer chain.
    if (!p.yield_issue) __builtin_amdgcn_s_setprio(3);

    // ---------------- stage 0 ----------------
    if constexpr (FUSED) {
        const size_t slab = (size_t)k * p.M * N;
        if constexpr (DBG) {
            correlate16<T, XT, true>(pXB + slab, pd + (size_t)k * p.M, p.M, sA, sr, lane, stamp, 8);
            stamp(10);
            correlate16<T, XT, false>(pXD + slab, (const XT*)nullptr, p.M, sB, sr, lane, stamp, 11);
        } else if constexpr (GS > 1) {
            // grouped spectra [K / GS][M L][GS]: the bin's elements are GS apart, starting at its place inside the group
            const size_t gslab = (size_t)(k / GS) * p.M * N * GS + (k % GS);
            correlate16<T, XT, true, NoStamp, GS>(pXB + gslab, pd + (size_t)k * p.M, p.M, sA, sr, lane);
            correlate16<T, XT, false, NoStamp, GS>(pXD + gslab, (const XT*)nullptr, p.M, sB, sr, lane);
        } else {
            correlate16<T, XT, true>(pXB + slab, pd + (size_t)k * p.M, p.M, sA, sr, lane);
            correlate16<T, XT, false>(pXD + slab, (const XT*)nullptr, p.M, sB, sr, lane);
        }
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
    stamp(1);
    if (dstop == 1) return;

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
            for (int t = 0; t < 4; ++t)
                if (4 * t <= kk) swr[buf][jq + 4 * t] = wrow[t];                // row kk of the working W (zero beyond column kk)
        }
        wsync();
        const T dkk = scol[buf][kk].x;
        if (!(dkk > (T)0) || !(dkk < (T)3.0e38)) { status = 1; break; }         // uniform: same LDS word for all lanes
        // 1/d for the updates now; the 1/sqrt(d) that scales row kk of W is taken once, for all rows together, after the loop
        const T inv2 = rcp_full(dkk);
        if (lane == 0) sPiv[kk] = dkk;
        const C li = scol[buf][i];
        const C li2 = mk<T>(li.x * inv2, li.y * inv2);
        // The outer-product update of B runs on the whole Hermitian matrix, without the triangle tests: rows and columns
        // already eliminated only cancel to rounding level and are never read again, and the unconditional update costs less
        // than its predicates.  W: rows past the pivot take the (unscaled) pivot row, which is zero beyond column kk.
        // What the unrolled loop knows at compile time is skipped: column groups of B entirely at or before the pivot
        // (4t + 3 <= kk) and groups of W entirely beyond it (4t > kk).
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (4 * t + 3 <= kk) continue;
            const C lj = scol[buf][jq + 4 * t];                  // B[i][j] -= B[i][kk] conj(B[j][kk]) / d
            // each component as two chained FMAs (written a -= p q + r s it compiles to a product, an FMA and a subtraction)
            brow[t].x = fma_t(-li2.y, lj.y, fma_t(-li2.x, lj.x, brow[t].x));
            brow[t].y = fma_t(li2.x, lj.y, fma_t(-li2.y, lj.x, brow[t].y));
        }
        if (i > kk) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (4 * t > kk) continue;
                const C wk = swr[buf][jq + 4 * t];               // W[i][j] -= (B[i][kk] / d) W[kk][j]
                wrow[t].x = fma_t(li2.y, wk.y, fma_t(-li2.x, wk.x, wrow[t].x));
                wrow[t].y = fma_t(-li2.y, wk.x, fma_t(-li2.x, wk.y, wrow[t].y));
            }
        }
    }
    wsync();
    if (status == 0) {
        // W = D^-1/2 (unit lower factor)^-1: every row by the reciprocal root of its pivot, one rsq for the sixteen rows at once
        // (inside the loop it was a serial rsq per step and a branch for the pivot row)
        const T ri = rsq_full(sPiv[i]);
#pragma unroll
        for (int t = 0; t < 4; ++t) wrow[t] = mk<T>(wrow[t].x * ri, wrow[t].y * ri);
    }
    stamp(2);
    if (dstop == 2) return;

    if (status == 0) {
        // W -> sB (R_D is spent)
#pragma unroll
        for (int t = 0; t < 4; ++t) sB[i * LD + jq + 4 * t] = wrow[t];
        wsync();
        // ---------------- stage 2: C = W A W^H ----------------
        C acc[4];
        const int col = lane & 15;
        cmm16([&](int r, int kx) { return sB[r * LD + kx]; }, [&](int kx, int c) { return sA[kx * LD + c]; }, lane, acc);
        wsync();
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
        // (wave-uniform: kept in scalar registers -- the vector registers of this kernel are all spoken for; as a vector value the
        // scaled norm was spilled on the common path, 8 bytes per lane = 17 MB of scratch written per launch)
        const T normF2 = uniform_scalar(wave_sum(nrm));
        wsync();
        stamp(3);
        if (dstop == 3) return;

        // ---------------- stage 3: register-resident Jacobi, XOR schedule ----------------
        const int a = lane >> 3, b = lane & 7;
        const int max_sweeps = p.max_sweeps > 0 ? p.max_sweeps : Prec<T>::max_sweeps;
        const T tol2 = p.sweep_tol2 > 0.0 ? (T)p.sweep_tol2 : Prec<T>::sweep_tol2;
        bool converged = false;
        // the scale exponent is wave-uniform: kept in a scalar register, the factors are re-formed where they are used
        const int sexp = __builtin_amdgcn_readfirstlane((normF2 > (T)0) ? -((sizeof(T) == 8 ? ilogb((double)normF2) : ilogbf((float)normF2)) / 2) : 0);
        T normS2;
        if constexpr (sizeof(T) == 8) normS2 = uniform_scalar((T)ldexp((double)normF2, 2 * sexp));
        else normS2 = uniform_scalar((T)ldexpf((float)normF2, 2 * sexp));
        bool v_in_lds = false, refined = false;
        // A one-sided solve is used only if it converged and its eigenvalues (the squared column norms) span less than 1e3: its
        // stop criterion is absolute, so columns of smaller norm (the null space of a rank-deficient C sits at the shift) may be
        // left askew.  Such a bin takes the two-sided sweeps on C, which is still intact in sA at that point.
        auto spectrum_ok = [&](float nt, float nb) {
            float mn = fminf(nt, nb), mx = fmaxf(nt, nb);
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {                                    // over the wave (any lane layout)
                mn = fminf(mn, __shfl_xor(mn, m, 64));
                mx = fmaxf(mx, __shfl_xor(mx, m, 64));
            }
            return !__any(!(mn >= 1e-3f * mx));                                   // NaN counts as not ok
        };
        if constexpr (sizeof(T) == 8) {
            // ---- float32 pre-solve (debug_stop == 4 skips it: double sweeps only, for A/B timing) -------------------
            // The sweeps are the cost of the kernel and packed-float ones cost a fraction of double ones, so C is first
            // diagonalised in float32: V32 with V32^H C V32 diagonal to ~1e-7.  V32 is then refined against the exact float64 C on
            // the matrix cores (below); W waits in registers meanwhile.
            if (dstop != 4) {
                using CF = Cx<float>;
                // looser than the float kernel's own 1e-8: the refinement follows anyway, so the float sweep that would only confirm
                // convergence is not run (1e-6 measured best: at 1e-5 so many more bins need a second refinement step that the launch
                // is 5 % slower, tools/probes/presolve_tol.py; debug_stop = 20 + e sets 1e-e)
                constexpr float kPresolveTol2 = 1e-6f;
                CF f0t, f0b, f1t, f1b;
                bool fconv = false, trust = true;
                int fs;
                int va = a, vb = b;                                                   // (row pair, slot) of this lane's part of V32
                if (dstop != 11) {
                    // one-sided form on the float Cholesky factor of 2^sexp C + delta I (same eigenvectors; the shift keeps the
                    // float pivots positive when C is singular to float precision).  The factor goes through sB, which is free
                    // until V32 lands there (W waits in registers), its column staging through the spent Cholesky staging.
                    constexpr int LDF = 17;
                    constexpr float kShift = 8e-6f;                                   // x ||C||_F (scaled to ~1)
                    CF* const fG = reinterpret_cast<CF*>(&sB[0]);
                    CF (*const fcol)[16] = reinterpret_cast<CF(*)[16]>(&scol[0][0]);
                    chol16_f32<T, LD, LDF>(sA, sexp, kShift * sqrtf((float)normS2), fG, fcol, lane);
                    stamp(4);
                    if (dstop == 12) return;                                   // timing aids: 12 after the float factor, 13 after the sweeps
                    va = lane & 7; vb = lane >> 3;                                   // the one-sided solve's layout: lane = a + 8 b
                    f0t = fG[(2 * va) * LDF + vb]; f0b = fG[(2 * va) * LDF + 8 + vb];
                    f1t = fG[(2 * va + 1) * LDF + vb]; f1b = fG[(2 * va + 1) * LDF + 8 + vb];
                    float n2t, n2b;
                    fs = jacobi16_onesided<LDF>(f0t, f0b, f1t, f1b, lane, (dstop >= 20 && dstop <= 27) ? __builtin_powif(10.f, 20 - dstop) : (dstop >= 30 && dstop <= 49) ? 0.5e-6f * (float)(dstop - 29) : kPresolveTol2, (float)normS2, Prec<float>::max_sweeps, fconv, n2t, n2b, fG);
                    trust = fconv && spectrum_ok(n2t, n2b);
                    stamp(5);
                    if (dstop == 13) {
                        if (lane == 0 && pstatus != nullptr) pstatus[k] = fs;        // sweeps of the pre-solve
                        return;
                    }
                } else {
                    // debug_stop == 11: round 2a's two-sided pre-solve (A/B timing)
                    auto ldf = [&](int r, int c) {
                        const C v = sA[r * LD + c];
                        return mk<float>((float)ldexp((double)v.x, sexp), (float)ldexp((double)v.y, sexp));
                    };
                    CF ftt = ldf(a, b), ftb = ldf(a, 8 + b), fbt = ldf(8 + a, b), fbb = ldf(8 + a, 8 + b);
                    f0t = mk<float>((2 * a == b) ? 1.f : 0.f, 0.f); f0b = mk<float>((2 * a == 8 + b) ? 1.f : 0.f, 0.f);
                    f1t = mk<float>((2 * a + 1 == b) ? 1.f : 0.f, 0.f); f1b = mk<float>((2 * a + 1 == 8 + b) ? 1.f : 0.f, 0.f);
                    fs = jacobi16_sweeps<float>(ftt, ftb, fbt, fbb, f0t, f0b, f1t, f1b, (float (*)[4]) nullptr, lane,
                                                kPresolveTol2, (float)normS2, Prec<float>::max_sweeps, fconv);
                }
                if (!trust) {
                    wsync();
#pragma unroll
                    for (int t = 0; t < 4; ++t) sB[i * LD + jq + 4 * t] = wrow[t];      // the float factor went through sB: W back in place
                    wsync();
                } else {
                // columns of V32 held by this lane: where the one-sided schedule leaves them, or (debug_stop == 11) the two-sided
                // schedules' layout after an odd / even number of sweeps
                const bool fnat = fs & 1;
                const int fit = (dstop != 11) ? os_top_end(vb) : (fnat ? 2 * vb : vb);
                const int fib = (dstop != 11) ? os_bot_end(vb) : (fnat ? 2 * vb + 1 : 8 + vb);
                const int mcol = lane & 15;
                auto cj = [](C w) { return mk<T>(w.x, -w.y); };
                wsync();
                sB[(2 * va) * LD + fit] = mk<T>((T)f0t.x, (T)f0t.y);                // V32 takes W's place
                sB[(2 * va) * LD + fib] = mk<T>((T)f0b.x, (T)f0b.y);
                sB[(2 * va + 1) * LD + fit] = mk<T>((T)f1t.x, (T)f1t.y);
                sB[(2 * va + 1) * LD + fib] = mk<T>((T)f1b.x, (T)f1b.y);
                wsync();
                C accT[4], accG[4], accC[4], accV[4];
                cmm16([&](int r, int kx) { return sA[r * LD + kx]; }, [&](int kx, int c) { return sB[kx * LD + c]; }, lane, accT);       // C V
                cmm16([&](int r, int kx) { return cj(sB[kx * LD + r]); }, [&](int kx, int c) { return sB[kx * LD + c]; }, lane, accG);   // V^H V
                wsync();
#pragma unroll
                for (int t = 0; t < 4; ++t) sA[mfma_row<T>(lane, t) * LD + mcol] = accT[t];
                wsync();
                cmm16([&](int r, int kx) { return cj(sB[kx * LD + r]); }, [&](int kx, int c) { return sA[kx * LD + c]; }, lane, accC);   // C1 = V^H C V
                // ---- one refinement step on the matrix cores in place of the double sweep (Ogita & Aishima 2018) ---------
                // With S = V^H C V, Gram = V^H V = I + E and the Rayleigh quotients d_i = S_ii / Gram_ii, the update
                //   V'' = V (I + Z),  Z_ij = (S_ij - d_j E_ij) / (d_j - d_i)  (i != j),  Z_ii = -E_ii / 2
                // removes the float solve's error to second order: what is left is |Z|^2, the same order a double Jacobi sweep
                // from (C', V') leaves.  |Z_ij| <= kRefineGuard on every pair keeps that below 1e-9; a wave that meets a
                // narrower spectral gap (or a pair the float sweeps did not finish) takes the double sweeps instead.
                constexpr double kRefineGuard2 = 9e-10;        // |Z_ij|^2 <= (3e-5)^2
                // A step that misses the guard by less than |Z_ij| <= 1e-2 is applied all the same and followed by a second one
                // in the rotated basis: S'' = (I+Z)^H S (I+Z) needs no C (the accumulator of S is itself an MFMA operand),
                // Gram'' = V''^H V''.  On the bench workload 24 % of the bins miss the guard at the first step (two eigenvalues
                // closer than ~3e-3 ||C||) and all but 0.1 % pass it at the second; the double sweeps are left for true clusters.
                constexpr double kSecondStep2 = 1e-4;
                // debug_stop == 5: always the double sweeps (A/B timing); a caller-set sweep tolerance (jdiag: 1e-17) asks for more
                // than the refinement's 1e-9 and gets the sweeps too
                const bool refine_ok = (dstop != 5) && !(p.sweep_tol2 > 0.0);
                // Z goes straight into sA (T = C V is spent: every lane is past the third product) and the second-order
                // eigenvalue terms into the idle coefficient array: nothing of this step stays in registers
                T* const sLam2 = reinterpret_cast<T*>(&scoef[0]);
                for (int step = 0; step < 2; ++step) {
                    {
                        // Rayleigh quotients S_jj / Gram_jj of the sixteen lanes that hold a diagonal element: the element is
                        // selected first and divided once, by reciprocal, Newton step and one residual correction (a double
                        // division per accumulator row under its own branch was ~150 instructions of this step)
                        T num = (T)0, gd = (T)1;
                        bool has_diag = false;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const bool d = mfma_row<T>(lane, t) == mcol;
                            num = d ? accC[t].x : num;
                            gd = d ? accG[t].x : gd;
                            has_diag = has_diag || d;
                        }
                        T ginv = __builtin_amdgcn_rcp(gd);
                        ginv = ginv * __builtin_fma(-gd, ginv, (T)2);
                        T quot = num * ginv;
                        quot = __builtin_fma(__builtin_fma(-gd, quot, num), ginv, quot);
                        if (has_diag) sLam[mcol] = quot;
                    }
                    wsync();
                    bool bad = false, hopeless = false;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int row = mfma_row<T>(lane, t);
                        const T di = sLam[row], dj = sLam[mcol];
                        T l2 = (T)0;
                        C z;
                        if (row == mcol) {
                            z = mk<T>((T)0.5 * ((T)1 - accG[t].x), (T)0);
                        } else {
                            const T den = dj - di;
                            T inv = __builtin_amdgcn_rcp(den);
                            inv = inv * __builtin_fma(-den, inv, (T)2);                  // one Newton step: full precision
                            const T zx = __builtin_fma(-dj, accG[t].x, accC[t].x) * inv, zy = __builtin_fma(-dj, accG[t].y, accC[t].y) * inv;
                            const T zz = zx * zx + zy * zy;
                            bad = bad || !(zz <= (T)kRefineGuard2);                      // NaN / inf (equal quotients) count as bad
                            hopeless = hopeless || !(zz <= (T)kSecondStep2);
                            z = mk<T>(zx, zy);
                            // second-order term of the eigenvalue of the pencil (S, Gram) next to d_i: -|S_ij - d_i E_ij|^2 / (d_j - d_i).
                            // (Exact to third order in Z AND E: the one-sided pre-solve leaves E ~ 1e-5 between the columns of small
                            // eigenvalues, where a formula that orthonormalises to first order only is off by E^2.)
                            const T nx = __builtin_fma(-di, accG[t].x, accC[t].x), ny = __builtin_fma(-di, accG[t].y, accC[t].y);
                            l2 = -(nx * nx + ny * ny) * inv;
                        }
                        sA[row * LD + mcol] = z;
                        // sum over the 16 lanes that share this row (lane ^ 1, 2, 4, 8), then add the quotient itself
                        l2 += xcol<1>(l2);
                        l2 += xcol<2>(l2);
                        l2 += xcol<4>(l2);
                        l2 += xrow<1>(l2, lane);
                        if (mcol == 0) sLam2[row] = l2 + di;
                    }
                    const bool pass = !__any(bad) || dstop == 10;                 // 10: guard off (timing aid, results invalid)
                    if (dstop == 9 && !pass) status = 8 << step;                  // 9: mark the bins by the step they miss
                    // the double sweeps start from the triple (V, S, Gram) as it is: leave before anything of it is touched
                    if (!refine_ok || (!pass && (step == 1 || __any(hopeless)))) break;
                    wsync();
                    auto v_step = [&]() {
                        cmm16([&](int r, int kx) { return sB[r * LD + kx]; }, [&](int kx, int c) { return sA[kx * LD + c]; }, lane, accV);   // V Z
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const C v = sB[mfma_row<T>(lane, t) * LD + mcol];
                            accV[t] = mk<T>(v.x + accV[t].x, v.y + accV[t].y);                                                               // V'' = V (I + Z)
                        }
                    };
                    if (pass) {
                        v_step();
                        if (lane < N) sLam[lane] = sLam2[lane];
                        wsync();
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            sA[mfma_row<T>(lane, t) * LD + mcol] = accV[t];             // Q for stage 5; sLam already holds the eigenvalues
                            sB[i * LD + jq + 4 * t] = wrow[t];                          // W back in place for stage 5
                        }
                        wsync();
                        refined = true;
                        break;
                    }
                    if constexpr (sizeof(T) == 8) {
                        // the triple after the step.  P = S (I + Z) takes S^T = conj(S) from its accumulator as the A operand,
                        // S'' = (I + Z)^H P takes P's accumulator as the B operand; then V'' and its Gram matrix
                        auto ipz = [&](int r, int c) { const C zv = sA[r * LD + c]; return mk<T>(zv.x + (r == c ? (T)1 : (T)0), zv.y); };
                        cmm16x([&](int s_, int, int) { return cj(accC[s_]); }, [&](int, int kx, int c) { return ipz(kx, c); }, lane, accT);           // P
                        cmm16x([&](int, int r, int kx) { return cj(ipz(kx, r)); }, [&](int s_, int, int) { return accT[s_]; }, lane, accC);          // S''
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (mfma_row<T>(lane, t) == mcol) accC[t].y = 0;
                        v_step();
                        wsync();
#pragma unroll
                        for (int t = 0; t < 4; ++t) sB[mfma_row<T>(lane, t) * LD + mcol] = accV[t];                                      // V''
                        wsync();
                        cmm16([&](int r, int kx) { return cj(sB[kx * LD + r]); }, [&](int kx, int c) { return sB[kx * LD + c]; }, lane, accG);   // Gram''
                    }
                }
                if (!refined) {
                    // ---- double sweeps instead ----------------------------------------------------------------------------------
                    // They start from an orthonormal V' and C' = V'^H C V'.  V' = V Y with Y = I - E/2 = (3 I - Gram)/2 is
                    // orthonormal to E^2 and C' = Y S Y exactly (Y is Hermitian; S and the intermediate product feed the MFMA
                    // from their accumulators).  A trusted pre-solve leaves |E| of a few 1e-2 at worst (one large eigenvalue and a cluster
                    // at 1.5e-3 of it: 3e-2 in the NumPy model), so four steps do with room to spare: 1e-1 -> 7.5e-3 -> 4e-5 -> 1e-9 -> 1e-18.
                    constexpr int n_it = 4;
                    {
                    for (int it = 0; it < n_it; ++it) {
                        wsync();
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int row = mfma_row<T>(lane, t);
                            sA[row * LD + mcol] = mk<T>((row == mcol ? (T)1.5 : (T)0) - (T)0.5 * accG[t].x, (T)-0.5 * accG[t].y);      // Y
                        }
                        wsync();
                        cmm16([&](int r, int kx) { return sB[r * LD + kx]; }, [&](int kx, int c) { return sA[kx * LD + c]; }, lane, accV);   // V Y
                        cmm16x([&](int s_, int, int) { return cj(accC[s_]); }, [&](int, int kx, int c) { return sA[kx * LD + c]; }, lane, accT);   // S Y
                        cmm16x([&](int, int r, int kx) { return sA[r * LD + kx]; }, [&](int s_, int, int) { return accT[s_]; }, lane, accC);       // Y (S Y)
                        wsync();
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            sB[mfma_row<T>(lane, t) * LD + mcol] = accV[t];                                                              // V'
                            if (mfma_row<T>(lane, t) == mcol) accC[t].y = 0;
                        }
                        wsync();
                        if (it + 1 < n_it)
                            cmm16([&](int r, int kx) { return cj(sB[kx * LD + r]); }, [&](int kx, int c) { return sB[kx * LD + c]; }, lane, accG);   // Gram'
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) sA[mfma_row<T>(lane, t) * LD + mcol] = accC[t];                                          // C'
                    wsync();
                    }
                    v_in_lds = true;
                }
                }
            }
        }
        if constexpr (sizeof(T) == 4) {
            // float kernel: the one-sided form IS the solve (debug_stop == 11: the two-sided sweeps below, for A/B timing).
            // Eigenvalues are the squared column norms less the shift, eigenvectors the normalised columns.
            if (dstop != 11) {
                constexpr int LDF = 17;
                constexpr float kShift = 8e-6f;
                const float delta = kShift * sqrtf((float)normS2);
                Cx<float>* const fG = reinterpret_cast<Cx<float>*>(&sB[0]);
                Cx<float> (*const fcol)[16] = reinterpret_cast<Cx<float>(*)[16]>(&scol[0][0]);
                chol16_f32<T, LD, LDF>(sA, sexp, delta, fG, fcol, lane);
                const int oa = lane & 7, ob = lane >> 3;                        // the one-sided solve's layout: lane = a + 8 b
                Cx<float> g0t = fG[(2 * oa) * LDF + ob], g0b = fG[(2 * oa) * LDF + 8 + ob];
                Cx<float> g1t = fG[(2 * oa + 1) * LDF + ob], g1b = fG[(2 * oa + 1) * LDF + 8 + ob];
                float n2t, n2b;
                (void)jacobi16_onesided<LDF>(g0t, g0b, g1t, g1b, lane, (float)tol2, (float)normS2, max_sweeps, converged, n2t, n2b, fG);
                const bool trust = converged && spectrum_ok(n2t, n2b);
                converged = false;
                wsync();
#pragma unroll
                for (int t = 0; t < 4; ++t) sB[i * LD + jq + 4 * t] = wrow[t];          // W back in place (stage 5, or the sweeps below)
                if (trust) {
                const int it_b = os_top_end(ob), ib_b = os_bot_end(ob);
                wsync();
                sA[(2 * oa) * LD + it_b] = mk<T>(g0t.x, g0t.y);
                sA[(2 * oa) * LD + ib_b] = mk<T>(g0b.x, g0b.y);
                sA[(2 * oa + 1) * LD + it_b] = mk<T>(g1t.x, g1t.y);
                sA[(2 * oa + 1) * LD + ib_b] = mk<T>(g1b.x, g1b.y);
                if (oa == 0) {
                    sLam[it_b] = (T)ldexpf(n2t - delta, -sexp);
                    sLam[ib_b] = (T)ldexpf(n2b - delta, -sexp);
                }
                refined = true;
                }
                wsync();
            }
        }
        if (!refined) {
            C tt = sA[a * LD + b], tb = sA[a * LD + 8 + b], bt = sA[(8 + a) * LD + b], bb = sA[(8 + a) * LD + 8 + b];
            const T scl = (sizeof(T) == 8) ? (T)ldexp(1.0, sexp) : (T)ldexpf(1.0f, sexp);
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
                const T iscl = (sizeof(T) == 8) ? (T)ldexp(1.0, -sexp) : (T)ldexpf(1.0f, -sexp);
                sLam[it_b] = tt.x * iscl;
                sLam[ib_b] = bb.x * iscl;
            }
            wsync();
        }

        stamp(6);
        // ---------------- stage 4: descending order ----------------
        // all 64 lanes: lane (i = lane & 15, q = lane >> 4) compares eigenvalue i with four others, the counts meet over q
        {
            const int i16 = lane & 15, q4 = lane >> 4;
            const T li = sLam[i16];
            int rank = 0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = 4 * q4 + jj;
                const T lj = sLam[j];
                rank += (lj > li) || (lj == li && j < i16);
            }
            rank += __shfl_xor(rank, 16, 64);
            rank += __shfl_xor(rank, 32, 64);
            if (q4 == 0) sOrder[rank] = i16;
        }

        // ---------------- stage 5: X = W^H Q ----------------
        cmm16([&](int r, int kx) { const C w = sB[kx * LD + r]; return mk<T>(w.x, -w.y); },
              [&](int kx, int c) { return sA[kx * LD + c]; }, lane, acc);
        wsync();
#pragma unroll
        for (int t = 0; t < 4; ++t) sA[mfma_row<T>(lane, t) * LD + col] = acc[t];
        wsync();

        // ---------------- stage 6: coefficients (x_i^H r) / (lam_i + mu) ----------------
        // all 64 lanes: lane (i, q) sums four of the sixteen terms of x_i^H r, the partial sums meet over q
        {
            const int i16 = lane & 15, q4 = lane >> 4;
            T sx = 0, sy = 0;
#pragma unroll
            for (int ll = 0; ll < 4; ++ll) {
                const int l = 4 * q4 + ll;
                const C v = sA[l * LD + i16], rr = sr[l];
                sx = fma_t(v.y, rr.y, fma_t(v.x, rr.x, sx));
                sy = fma_t(-v.y, rr.x, fma_t(v.x, rr.y, sy));
            }
            sx += __shfl_xor(sx, 16, 64); sy += __shfl_xor(sy, 16, 64);
            sx += __shfl_xor(sx, 32, 64); sy += __shfl_xor(sy, 32, 64);
            if (q4 == 0) {
                const T den = rcp_full(sLam[i16] + (T)p.mu);
                scoef[i16] = mk<T>(sx * den, sy * den);
            }
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
                    ax = fma_t(-cf.y, v.y, fma_t(cf.x, v.x, ax));
                    ay = fma_t(cf.y, v.x, fma_t(cf.x, v.y, ay));
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
    stamp(7);
}

// The double kernel is held to four waves per SIMD (its float32 pre-solve and the re-orthonormalisation products would
// otherwise raise the register count past 128 and cost a wave); the float kernel is left to the compiler.
template <typename T, bool FUSED, typename XT, bool DBG = false>
__global__ void __launch_bounds__(64) gevd16m_kernel(const GevdParams p) { gevd16m_body<T, FUSED, XT, DBG>(p, blockIdx.x, blockIdx.y == 1); }
template <bool FUSED, typename XT, bool DBG = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) gevd16m_kernel_f64(const GevdParams p) {
    gevd16m_body<double, FUSED, XT, DBG>(p, blockIdx.x, blockIdx.y == 1);
}
// The float64 streaming front-end's grouped spectra (p.x_group = GS).  Workgroups go to the eight XCDs round robin by their linear
// index and the GS bins of a group read the same lines: index 8 GS q + 8 r + x (x = XCD, r < GS) takes bin 8 GS q + GS x + r, so
// that a group's bins share one L2 (the last, partial block of 8 GS keeps its order).
template <int GS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) gevd16m_kernel_f64_grouped(const GevdParams p) {
    constexpr int BLK = 8 * GS;
    const int id = blockIdx.x;
    const int k = ((id | (BLK - 1)) < p.K) ? ((id & ~(BLK - 1)) | ((id & 7) * GS) | ((id >> 3) & (GS - 1))) : id;
    gevd16m_body<double, true, double2, false, GS>(p, k, blockIdx.y == 1);
}
// The hops of a chunk in one launch (chunked whole-signal path: blockIdx.z = hop), bin-major (GS = 1) or grouped spectra.  A hop's 2050
// waves are two per SIMD -- all head and tail; sixteen hops are a launch of the headline's size.
template <int GS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) gevd16m_kernel_f64_hops(const GevdParams p) {
    constexpr int BLK = 8 * GS;
    const int id = blockIdx.x;
    const int k = (GS > 1 && (id | (BLK - 1)) < p.K) ? ((id & ~(BLK - 1)) | ((id & 7) * GS) | ((id >> 3) & (GS - 1))) : id;
    gevd16m_body<double, true, double2, false, GS, true>(p, k, blockIdx.y == 1, blockIdx.z);
}

// the same for c64 slabs (float32 front end): float64 arithmetic ("mixed") and float32 arithmetic
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) gevd16m_kernel_f64_hops_c64(const GevdParams p) {
    gevd16m_body<double, true, float2, false, 1, true>(p, blockIdx.x, blockIdx.y == 1, blockIdx.z);
}
__global__ void __launch_bounds__(64) gevd16m_kernel_f32_hops_c64(const GevdParams p) {
    gevd16m_body<float, true, float2, false, 1, true>(p, blockIdx.x, blockIdx.y == 1, blockIdx.z);
}

}  // namespace

bool apv_gevd16m_takes_hops(const GevdParams& p, int compute_dtype, bool fused) {
    if (!(p.n == 16 && p.reg_mode == APV_REG_ABS && p.reg_bright == 0.0 && fused && p.debug_stop == 0 && p.stamps == nullptr && p.n_hops <= 65535))
        return false;
    if (p.x_c128) return compute_dtype == APV_F64 && (p.x_group <= 1 || p.x_group == 4 || p.x_group == 8);
    return p.x_group <= 1;          // c64 slabs: either arithmetic, bin-major
}

hipError_t apv_launch_gevd16m(const GevdParams& p, int compute_dtype, bool fused, hipStream_t s) {
    if (p.n != 16 || p.reg_mode != APV_REG_ABS || p.reg_bright != 0.0) return hipErrorNotSupported;
    if (p.K <= 0) return hipSuccess;
    const dim3 grid(p.K, p.n_zones > 1 ? 2 : 1);
    const bool xd = fused && p.x_c128;
    if (p.n_hops > 1) {
        if (!apv_gevd16m_takes_hops(p, compute_dtype, fused)) return hipErrorInvalidValue;
        const dim3 grid3(p.K, p.n_zones > 1 ? 2 : 1, p.n_hops);
        if (!p.x_c128) {
            if (compute_dtype == APV_F64) hipLaunchKernelGGL(gevd16m_kernel_f64_hops_c64, grid3, dim3(64), 0, s, p);
            else hipLaunchKernelGGL(gevd16m_kernel_f32_hops_c64, grid3, dim3(64), 0, s, p);
            return hipGetLastError();
        }
        if (p.x_group == 4) hipLaunchKernelGGL(gevd16m_kernel_f64_hops<4>, grid3, dim3(64), 0, s, p);
        else if (p.x_group == 8) hipLaunchKernelGGL(gevd16m_kernel_f64_hops<8>, grid3, dim3(64), 0, s, p);
        else hipLaunchKernelGGL(gevd16m_kernel_f64_hops<1>, grid3, dim3(64), 0, s, p);
        return hipGetLastError();
    }
    if (p.x_group > 1) {
        // only the float64 product kernel on c128 slabs reads the grouped layout (apv_gevd16m_reads_groups says when)
        if (!xd || compute_dtype != APV_F64 || p.debug_stop != 0 || p.stamps != nullptr) return hipErrorInvalidValue;
        if (p.x_group == 4) hipLaunchKernelGGL(gevd16m_kernel_f64_grouped<4>, grid, dim3(64), 0, s, p);
        else if (p.x_group == 8) hipLaunchKernelGGL(gevd16m_kernel_f64_grouped<8>, grid, dim3(64), 0, s, p);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (p.debug_stop != 0 || p.stamps != nullptr) {
        // diagnostic instantiations (probes only): stage cuts, A/B switches, stage stamps
        if (compute_dtype == APV_F64) {
            if (xd) hipLaunchKernelGGL((gevd16m_kernel_f64<true, double2, true>), grid, dim3(64), 0, s, p);
            else if (fused) hipLaunchKernelGGL((gevd16m_kernel_f64<true, float2, true>), grid, dim3(64), 0, s, p);
            else hipLaunchKernelGGL((gevd16m_kernel_f64<false, float2, true>), grid, dim3(64), 0, s, p);
        } else {
            if (xd) hipLaunchKernelGGL((gevd16m_kernel<float, true, double2, true>), grid, dim3(64), 0, s, p);
            else if (fused) hipLaunchKernelGGL((gevd16m_kernel<float, true, float2, true>), grid, dim3(64), 0, s, p);
            else hipLaunchKernelGGL((gevd16m_kernel<float, false, float2, true>), grid, dim3(64), 0, s, p);
        }
        return hipGetLastError();
    }
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
