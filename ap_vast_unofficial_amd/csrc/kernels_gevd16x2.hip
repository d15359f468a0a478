// Order-16 fused subband update in float64: workgroups of two waves and two frequency bins, the float pre-solve of both bins in ONE wave.
//
// Why.  On gfx950 the 16 x 16 x 4 MFMAs (f64 and f32 alike) issue through the SAME pipe as the vector instructions of the other
// waves of the SIMD (profiles/r03/coissue.md: matrix loop + vector loop = the SUM of their times, not the maximum), and at four
// waves per SIMD the one-bin-per-wave kernel (kernels_gevd16m.hip) keeps that pipe busy: its time is its instruction count.  62 %
// of its vector instructions are the float32 one-sided Jacobi sweeps, where all 64 lanes work on ONE 16 x 16 matrix: eight lanes
// share a column pair and each of them derives the pair's rotation (20 of the 51 instructions of a round, three of them
// transcendental).  With two bins in the sweeping wave a column pair is shared by FOUR lanes (rows 4a .. 4a+3 each): a round is 62
// instructions for two bins instead of 51 for one.  Everything else -- correlation, Cholesky + inverse, whitening, the float factor,
// the refinement on the matrix cores, sort, back-transform, filter -- is the one-bin code, one wave per bin, the two waves of a
// workgroup side by side.  Per bin: ~4 400 vector instructions instead of ~5 600, the same 134 MFMAs.
//
// (First form tried: ONE wave carrying both bins through every stage, bin 0 then bin 1.  168 VGPRs = 3 waves per SIMD and a
// dependent chain twice as long per wave: latency bound, 0.63 ms per 32 768 bins against 0.49 ms for the one-bin kernel.)
//
// Scope.  The common path only: float pre-solve, one or two refinement steps.  A bin whose pre-solve cannot be trusted (eigenvalue
// span, no convergence) or whose refinement misses its guard twice is appended to a redo list; the one-bin kernel
// (gevd16m_kernel_f64, LIST form) then recomputes exactly those bins with its double sweeps (one bin in 30 000 on the bench
// workload).  A dark matrix that is not positive definite is reported here (status 1), as in the one-bin kernel.
//
//   stages and reference anchors: see kernels_gevd16m.hip (apvast.py:20-36, 329-364, 406-414)
#include "apv_internal.h"

#include <cstdlib>

// in this translation unit the LDS ordering points of the shared device helpers are wave-local: a workgroup is two waves that go
// their own ways between two explicit barriers (see gevd16x2_kernel)
#define APV_WSYNC_WAVE_LOCAL 1
#include "gevd16_common.h"

namespace {

using CF = Cx<float>;
using CD = Cx<double>;

// ---- the joint one-sided sweeps: lane = a + 4 b + 32 bin; a: rows 4a .. 4a+3, b: column slot (top, bottom) -------------------------
// Same schedule, same start and end slots as jacobi16_onesided (gevd16_common.h): only the lane layout differs.
__device__ __forceinline__ float half_sum(float v) {        // over the 32 lanes of a bin's half of the wave
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over the four lanes that share a column slot (lane bits 0, 1): two DPP adds on the VALU
__device__ __forceinline__ float quadsum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
    return v;
}
template <int XORMASK> __device__ __forceinline__ float swz(float v) {     // lane ^ XORMASK within 32 lanes, on the LDS crossbar
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x1F | (XORMASK << 10)));
}
template <int XORMASK> __device__ __forceinline__ CF cswz(CF v) { return mk<float>(swz<XORMASK>(v.x), swz<XORMASK>(v.y)); }

// stage: per bin a [16][LDF] float staging in LDS through which the columns go back to their starting slots between two sweeps
template <int LDF>
__device__ __forceinline__ void jacobi16_onesided_x2(CF (&top)[4], CF (&bot)[4], int lane, float tol2, float normS2, int max_sweeps,
                                                     bool& converged_, float& n2t_, float& n2b_, CF* stage) {
    const int a = lane & 3, b = (lane >> 2) & 7;
    bool converged = false;                                   // of this lane's bin
    CF t0 = top[0], t1 = top[1], t2 = top[2], t3 = top[3], b0 = bot[0], b1 = bot[1], b2 = bot[2], b3 = bot[3];
    auto norm2 = [&](CF x0, CF x1, CF x2, CF x3) {
        return quadsum(x0.x * x0.x + x0.y * x0.y + x1.x * x1.x + x1.y * x1.y + x2.x * x2.x + x2.y * x2.y + x3.x * x3.x + x3.y * x3.y);
    };
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        // (both bins run every sweep until both have converged or the cap is reached: a converged bin's further rotations are
        // tiny and harmless; the wave cannot skip half of itself)
        if (sweep > 0) {
            wsync();
            const int ct = os_top_end(b), cb = os_bot_end(b);
            stage[(4 * a + 0) * LDF + ct] = t0; stage[(4 * a + 1) * LDF + ct] = t1; stage[(4 * a + 2) * LDF + ct] = t2; stage[(4 * a + 3) * LDF + ct] = t3;
            stage[(4 * a + 0) * LDF + cb] = b0; stage[(4 * a + 1) * LDF + cb] = b1; stage[(4 * a + 2) * LDF + cb] = b2; stage[(4 * a + 3) * LDF + cb] = b3;
            wsync();
            t0 = stage[(4 * a + 0) * LDF + b]; t1 = stage[(4 * a + 1) * LDF + b]; t2 = stage[(4 * a + 2) * LDF + b]; t3 = stage[(4 * a + 3) * LDF + b];
            b0 = stage[(4 * a + 0) * LDF + 8 + b]; b1 = stage[(4 * a + 1) * LDF + 8 + b]; b2 = stage[(4 * a + 2) * LDF + 8 + b]; b3 = stage[(4 * a + 3) * LDF + 8 + b];
        }
        float off = 0.f;
        float nt = norm2(t0, t1, t2, t3), nb = norm2(b0, b1, b2, b3);
#pragma unroll
        for (int r = 0; r < 15; ++r) {
            const int delta = (int)((OS_DELTA >> (4 * r)) & 15), tbit = (int)((OS_TBIT >> (4 * r)) & 15) - 1;
            // re-deal of tops and bottoms between slots b and b ^ (1 << tbit), three times per sweep: the lanes whose slot bit is
            // set give their top and take the partner's bottom (lane ^ (4 << tbit), within the bin's 32 lanes)
            if (tbit >= 0) {
                const bool bit = (b >> tbit) & 1;
                auto ex = [&](float& tp, float& bt) {
                    const float send = bit ? tp : bt;
                    float recv;
                    if (tbit == 0) recv = swz<4>(send);
                    else if (tbit == 1) recv = swz<8>(send);
                    else recv = swz<16>(send);
                    if (bit) tp = recv; else bt = recv;
                };
                ex(t0.x, b0.x); ex(t0.y, b0.y); ex(t1.x, b1.x); ex(t1.y, b1.y);
                ex(t2.x, b2.x); ex(t2.y, b2.y); ex(t3.x, b3.x); ex(t3.y, b3.y);
                ex(nt, nb);
            }
            // the bottoms move by slot-XOR delta = lane ^ (4 delta): nine crossbar moves (compile-time patterns)
            if (delta == 1) { b0 = cswz<4>(b0); b1 = cswz<4>(b1); b2 = cswz<4>(b2); b3 = cswz<4>(b3); nb = swz<4>(nb); }
            else if (delta == 2) { b0 = cswz<8>(b0); b1 = cswz<8>(b1); b2 = cswz<8>(b2); b3 = cswz<8>(b3); nb = swz<8>(nb); }
            else if (delta == 3) { b0 = cswz<12>(b0); b1 = cswz<12>(b1); b2 = cswz<12>(b2); b3 = cswz<12>(b3); nb = swz<12>(nb); }
            else if (delta == 7) { b0 = cswz<28>(b0); b1 = cswz<28>(b1); b2 = cswz<28>(b2); b3 = cswz<28>(b3); nb = swz<28>(nb); }
            // pivot beta = g_top^H g_bottom over the 16 rows: this lane's four, then the four lanes of the slot
            f2v pa = (f2v){t0.x, t0.x} * (f2v){b0.x, b0.y};
            pa = __builtin_elementwise_fma((f2v){t1.x, t1.x}, (f2v){b1.x, b1.y}, pa);
            pa = __builtin_elementwise_fma((f2v){t2.x, t2.x}, (f2v){b2.x, b2.y}, pa);
            pa = __builtin_elementwise_fma((f2v){t3.x, t3.x}, (f2v){b3.x, b3.y}, pa);
            f2v pb = (f2v){t0.y, t0.y} * (f2v){b0.y, b0.x};
            pb = __builtin_elementwise_fma((f2v){t1.y, t1.y}, (f2v){b1.y, b1.x}, pb);
            pb = __builtin_elementwise_fma((f2v){t2.y, t2.y}, (f2v){b2.y, b2.x}, pb);
            pb = __builtin_elementwise_fma((f2v){t3.y, t3.y}, (f2v){b3.y, b3.x}, pb);
            float bx = pa.x + pb.x, by = pa.y - pb.y;
#define APV_DPP_ADD(v, ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, true))
            APV_DPP_ADD(bx, 0xB1); asm volatile("" : "+v"(bx)); APV_DPP_ADD(by, 0xB1); asm volatile("" : "+v"(by));
            APV_DPP_ADD(bx, 0x4E); asm volatile("" : "+v"(bx)); APV_DPP_ADD(by, 0x4E);
#undef APV_DPP_ADD
            const float bb2 = bx * bx + by * by;
            off += bb2;
            // rotation for [[nt, beta], [conj(beta), nb]] (see jacobi16_onesided)
            const float zeta = 0.5f * (nb - nt);
            const float x = fmaxf(__builtin_fmaf(zeta, zeta, bb2), 1e-36f);
            const float D = fabsf(zeta) + x * __builtin_amdgcn_rsqf(x);
            const float tsgn = copysignf(__builtin_amdgcn_rcpf(D), zeta);
            const float tx = bx * tsgn, ty = by * tsgn;
            const float c = __builtin_amdgcn_rsqf(__builtin_fmaf(tx, tx, __builtin_fmaf(ty, ty, 1.0f)));
            const CF s = mk<float>(tx * c, ty * c);
            const float shift = bb2 * tsgn;
            nt -= shift;
            nb += shift;
            CF wp, wq;
            rot_cols<float>(c, s, t0, b0, wp, wq); t0 = wp; b0 = wq;
            rot_cols<float>(c, s, t1, b1, wp, wq); t1 = wp; b1 = wq;
            rot_cols<float>(c, s, t2, b2, wp, wq); t2 = wp; b2 = wq;
            rot_cols<float>(c, s, t3, b3, wp, wq); t3 = wp; b3 = wq;
        }
        // every pair was counted by the four lanes of its slot
        const float tot = half_sum(off) * 0.25f;
        converged = tot <= tol2 * normS2;
        if (!__any(!converged)) break;                       // both bins done
    }
    n2t_ = norm2(t0, t1, t2, t3);
    n2b_ = norm2(b0, b1, b2, b3);
    const float it = __builtin_amdgcn_rsqf(fmaxf(n2t_, 1e-30f)), ib = __builtin_amdgcn_rsqf(fmaxf(n2b_, 1e-30f));
    top[0] = mk<float>(t0.x * it, t0.y * it); top[1] = mk<float>(t1.x * it, t1.y * it);
    top[2] = mk<float>(t2.x * it, t2.y * it); top[3] = mk<float>(t3.x * it, t3.y * it);
    bot[0] = mk<float>(b0.x * ib, b0.y * ib); bot[1] = mk<float>(b1.x * ib, b1.y * ib);
    bot[2] = mk<float>(b2.x * ib, b2.y * ib); bot[3] = mk<float>(b3.x * ib, b3.y * ib);
    converged_ = converged;
}

// what stages 0 - 3a leave behind for one bin
struct Bin {
    CD wrow[4];         // W = L^-1: lane (i = lane >> 2, jq = lane & 3) holds W[i][jq + 4t]
    double rx, ry;      // r = X_B^H d: lane l < 16 holds r[l]
    double normS2;      // ||2^sexp C||_F^2
    int sexp;
    int status;         // 0 | 1 not positive definite | 3 redo (one-bin kernel)
};

// ---- stages 0 - 3a of bin k: C -> sC, the float Cholesky factor of 2^sexp C + delta I -> fG (in sW), the rest in `o` ----------------
template <typename XT>
__device__ __forceinline__ void front_half(const GevdParams& p, const XT* pXB, const XT* pXD, const XT* pd, int k, CD* sC, CD* sW,
                                           Bin& o, int lane) {
    using T = double;
    using C = CD;
    constexpr int LDF = 17;
    // the Cholesky staging and the float factor's column staging live inside sW while it holds nothing else
    C (*const scol)[N] = reinterpret_cast<C(*)[N]>(sW);                     // [2][N]
    C (*const swr)[N] = reinterpret_cast<C(*)[N]>(sW + 2 * N);              // [2][N]
    T* const sPiv = reinterpret_cast<T*>(sW + 4 * N);                       // [N]
    C* const srtmp = sW + 5 * N;                                            // [N]: correlate16 writes r here
    o.status = 0;
    // ---------------- stage 0 ----------------
    const size_t slab = (size_t)k * p.M * N;
    // R_D first: it goes to registers at once and frees sW for r's staging
    correlate16<T, XT>(pXD + slab, (const XT*)nullptr, p.M, sW, srtmp, lane);
    wsync();
    const int i = lane >> 2, jq = lane & 3;
    C brow[4], wrow[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = jq + 4 * t;
        brow[t] = sW[i * LD + j];
        if (j == i) brow[t] = mk<T>(brow[t].x + (T)p.reg_dark, 0);
        wrow[t] = mk<T>((j == i) ? (T)1 : (T)0, 0);
    }
    wsync();
    correlate16<T, XT>(pXB + slab, pd + (size_t)k * p.M, p.M, sC, srtmp, lane);
    wsync();
    {
        const C rv = srtmp[lane & 15];
        o.rx = rv.x;
        o.ry = rv.y;
    }
    wsync();
    // ---------------- stage 1: Cholesky of B + reg I with W = L^-1 (kernels_gevd16m.hip) ----------------
#pragma unroll
    for (int kk = 0; kk < N; ++kk) {
        const int buf = kk & 1;
        if (jq == (kk & 3)) scol[buf][i] = brow[kk >> 2];
        if (i == kk) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (4 * t <= kk) swr[buf][jq + 4 * t] = wrow[t];
        }
        wsync();
        const T dkk = scol[buf][kk].x;
        if (!(dkk > (T)0) || !(dkk < (T)3.0e38)) { o.status = 1; break; }
        const T inv2 = rcp_full(dkk);
        if (lane == 0) sPiv[kk] = dkk;
        const C li = scol[buf][i];
        const C li2 = mk<T>(li.x * inv2, li.y * inv2);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (4 * t + 3 <= kk) continue;
            const C lj = scol[buf][jq + 4 * t];
            brow[t].x = fma_t(-li2.y, lj.y, fma_t(-li2.x, lj.x, brow[t].x));
            brow[t].y = fma_t(li2.x, lj.y, fma_t(-li2.y, lj.x, brow[t].y));
        }
        if (i > kk) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (4 * t > kk) continue;
                const C wk = swr[buf][jq + 4 * t];
                wrow[t].x = fma_t(li2.y, wk.y, fma_t(-li2.x, wk.x, wrow[t].x));
                wrow[t].y = fma_t(-li2.y, wk.x, fma_t(-li2.x, wk.y, wrow[t].y));
            }
        }
    }
    wsync();
    if (o.status != 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) o.wrow[t] = mk<T>(0, 0);
        o.normS2 = 0;
        o.sexp = 0;
        return;
    }
    {
        const T ri = rsq_full(sPiv[i]);
#pragma unroll
        for (int t = 0; t < 4; ++t) wrow[t] = mk<T>(wrow[t].x * ri, wrow[t].y * ri);
    }
    wsync();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        sW[i * LD + jq + 4 * t] = wrow[t];
        o.wrow[t] = wrow[t];
    }
    wsync();
    // ---------------- stage 2: C = W A W^H ----------------
    C acc[4];
    const int col = lane & 15;
    cmm16([&](int r, int kx) { return sW[r * LD + kx]; }, [&](int kx, int c) { return sC[kx * LD + c]; }, lane, acc);
    wsync();
#pragma unroll
    for (int t = 0; t < 4; ++t) sC[mfma_row<T>(lane, t) * LD + col] = acc[t];
    wsync();
    cmm16([&](int r, int kx) { return sC[r * LD + kx]; },
          [&](int kx, int c) { const C w = sW[c * LD + kx]; return mk<T>(w.x, -w.y); }, lane, acc);
    wsync();
    T nrm = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = mfma_row<T>(lane, t);
        if (row == col) acc[t].y = 0;
        sC[row * LD + col] = acc[t];
        nrm += acc[t].x * acc[t].x + acc[t].y * acc[t].y;
    }
    const T normF2 = wave_sum(nrm);
    wsync();
    o.sexp = __builtin_amdgcn_readfirstlane((normF2 > (T)0) ? -(ilogb((double)normF2) / 2) : 0);
    o.normS2 = ldexp((double)normF2, 2 * o.sexp);
    // ---------------- stage 3a: float Cholesky factor of 2^sexp C + delta I -> fG (sW: W lives in registers now) ----------------
    constexpr float kShift = 8e-6f;
    CF* const fG = reinterpret_cast<CF*>(sW);                               // [16][LDF] floats: 2 176 B
    CF (*const fcol)[16] = reinterpret_cast<CF(*)[16]>(reinterpret_cast<char*>(sW) + 2304);   // [2][16] behind it
    chol16_f32<T, LD, LDF>(sC, o.sexp, kShift * sqrtf((float)o.normS2), fG, fcol, lane);
}

// ---- stages 3c - 6 and the outputs of bin k; sW holds its float eigenvectors V32 (as float64) ------------------------------------
// returns the status: 0, or 3 when the refinement could not certify the pre-solve (redo)
__device__ __forceinline__ int back_half(const GevdParams& p, void* pw, void* plam, int k, CD* sC, CD* sW, double* sLam, double* sLam2,
                                         int* sOrder, const Bin& bin, int lane) {
    using T = double;
    using C = CD;
    const int i = lane >> 2, jq = lane & 3;
    const int mcol = lane & 15;
    auto cj = [](C w) { return mk<T>(w.x, -w.y); };
    C accT[4], accG[4], accC[4], accV[4];
    cmm16([&](int r, int kx) { return sC[r * LD + kx]; }, [&](int kx, int c) { return sW[kx * LD + c]; }, lane, accT);       // C V
    cmm16([&](int r, int kx) { return cj(sW[kx * LD + r]); }, [&](int kx, int c) { return sW[kx * LD + c]; }, lane, accG);   // V^H V
    wsync();
#pragma unroll
    for (int t = 0; t < 4; ++t) sC[mfma_row<T>(lane, t) * LD + mcol] = accT[t];
    wsync();
    cmm16([&](int r, int kx) { return cj(sW[kx * LD + r]); }, [&](int kx, int c) { return sC[kx * LD + c]; }, lane, accC);   // S = V^H C V
    // refinement steps on the matrix cores (Ogita & Aishima 2018): see kernels_gevd16m.hip, stage 3
    constexpr double kRefineGuard2 = 9e-10, kSecondStep2 = 1e-4;
    bool refined = false;
    for (int step = 0; step < 2; ++step) {
        {
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
                inv = inv * __builtin_fma(-den, inv, (T)2);
                const T zx = __builtin_fma(-dj, accG[t].x, accC[t].x) * inv, zy = __builtin_fma(-dj, accG[t].y, accC[t].y) * inv;
                const T zz = zx * zx + zy * zy;
                bad = bad || !(zz <= (T)kRefineGuard2);
                hopeless = hopeless || !(zz <= (T)kSecondStep2);
                z = mk<T>(zx, zy);
                const T nx = __builtin_fma(-di, accG[t].x, accC[t].x), ny = __builtin_fma(-di, accG[t].y, accC[t].y);
                l2 = -(nx * nx + ny * ny) * inv;
            }
            sC[row * LD + mcol] = z;
            l2 += xcol<1>(l2);
            l2 += xcol<2>(l2);
            l2 += xcol<4>(l2);
            l2 += xrow<1>(l2, lane);
            if (mcol == 0) sLam2[row] = l2 + di;
        }
        const bool pass = !__any(bad);
        if (!pass && (step == 1 || __any(hopeless))) break;
        wsync();
        auto v_step = [&]() {
            cmm16([&](int r, int kx) { return sW[r * LD + kx]; }, [&](int kx, int c) { return sC[kx * LD + c]; }, lane, accV);   // V Z
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const C v = sW[mfma_row<T>(lane, t) * LD + mcol];
                accV[t] = mk<T>(v.x + accV[t].x, v.y + accV[t].y);
            }
        };
        if (pass) {
            v_step();
            if (lane < N) sLam[lane] = sLam2[lane];
            wsync();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                sC[mfma_row<T>(lane, t) * LD + mcol] = accV[t];                 // Q
                sW[i * LD + jq + 4 * t] = bin.wrow[t];                          // W back in place for stage 5
            }
            wsync();
            refined = true;
            break;
        }
        // a second step in the rotated basis: S'' = (I + Z)^H S (I + Z), V'' = V (I + Z), Gram'' = V''^H V''
        auto ipz = [&](int r, int c) { const C zv = sC[r * LD + c]; return mk<T>(zv.x + (r == c ? (T)1 : (T)0), zv.y); };
        cmm16x([&](int s_, int, int) { return cj(accC[s_]); }, [&](int, int kx, int c) { return ipz(kx, c); }, lane, accT);
        cmm16x([&](int, int r, int kx) { return cj(ipz(kx, r)); }, [&](int s_, int, int) { return accT[s_]; }, lane, accC);
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (mfma_row<T>(lane, t) == mcol) accC[t].y = 0;
        v_step();
        wsync();
#pragma unroll
        for (int t = 0; t < 4; ++t) sW[mfma_row<T>(lane, t) * LD + mcol] = accV[t];
        wsync();
        cmm16([&](int r, int kx) { return cj(sW[kx * LD + r]); }, [&](int kx, int c) { return sW[kx * LD + c]; }, lane, accG);
    }
    if (!refined) return 3;
    // ---------------- stage 4: descending order ----------------
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
    C acc[4];
    cmm16([&](int r, int kx) { const C w = sW[kx * LD + r]; return mk<T>(w.x, -w.y); }, [&](int kx, int c) { return sC[kx * LD + c]; }, lane, acc);
    wsync();
#pragma unroll
    for (int t = 0; t < 4; ++t) sC[mfma_row<T>(lane, t) * LD + mcol] = acc[t];
    // W is spent: r and the coefficients take its place
    C* const sr = sW;                  // [N]
    C* const scoef = sW + N;           // [N]
    if (lane < N) sr[lane] = mk<T>(bin.rx, bin.ry);
    wsync();
    // ---------------- stage 6: coefficients (x_i^H r) / (lam_i + mu) ----------------
    {
        const int i16 = lane & 15, q4 = lane >> 4;
        T sx = 0, sy = 0;
#pragma unroll
        for (int ll = 0; ll < 4; ++ll) {
            const int l = 4 * q4 + ll;
            const C v = sC[l * LD + i16], rr = sr[l];
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
    // ---------------- outputs ----------------
    if (lane < N) {
        T ax = 0, ay = 0;
        int done = 0;
        for (int t = 0; t < p.nV; ++t) {
            const int V = p.ranks[t];
            for (; done < V; ++done) {
                const int c = sOrder[done];
                const C cf = scoef[c], v = sC[lane * LD + c];
                ax = fma_t(-cf.y, v.y, fma_t(cf.x, v.x, ax));
                ay = fma_t(cf.y, v.x, fma_t(cf.x, v.y, ay));
            }
            const size_t oi = ((size_t)k * p.nV + t) * N + lane;
            if (p.out_c128) reinterpret_cast<double2*>(pw)[oi] = make_double2(ax, ay);
            else reinterpret_cast<float2*>(pw)[oi] = make_float2((float)ax, (float)ay);
        }
        if (plam != nullptr) {
            const T lv = sLam[sOrder[lane]];
            if (p.out_c128) reinterpret_cast<double*>(plam)[(size_t)k * N + lane] = lv;
            else reinterpret_cast<float*>(plam)[(size_t)k * N + lane] = (float)lv;
        }
    }
    wsync();
    return 0;
}

// A workgroup is TWO waves and two bins.  Wave w runs the front half of bin w (its own C and work matrix in LDS); the workgroup
// meets; wave 0 runs the float pre-solve of BOTH bins (32 lanes each) while wave 1 sleeps at the second barrier (a sleeping wave
// takes no issue slot: that is the point); then wave w runs the back half of bin w.  Inside the halves the LDS ordering points are
// wave-local (APV_WSYNC_WAVE_LOCAL above): the two waves go their own ways -- a bin that is not positive definite, refinement
// steps -- and only the two s_barriers below are common.
template <typename XT>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) gevd16x2_kernel(const GevdParams p) {
    constexpr int LDF = 17;
    __shared__ CD sC[2][N * LD];       // per bin: R_B -> W R_B -> C -> (refinement) Z -> Q -> X
    __shared__ CD sW[2][N * LD];       // per bin: R_D / staging -> W -> float factor, sweep staging -> V -> W -> r, coefficients
    __shared__ double sLam[2][N], sLam2[2][N];
    __shared__ int sOrder[2][N];
    __shared__ int sTrust[2];
    const bool z1 = (blockIdx.y == 1);
    const XT* const pXB = reinterpret_cast<const XT*>(z1 ? p.XB1 : p.XB);
    const XT* const pXD = reinterpret_cast<const XT*>(z1 ? p.XD1 : p.XD);
    const XT* const pd = reinterpret_cast<const XT*>(z1 ? p.d1 : p.d);
    void* const pw = z1 ? p.w1 : p.w;
    void* const plam = z1 ? p.lam1 : p.lam;
    int32_t* const pstatus = z1 ? p.status1 : p.status;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = 2 * blockIdx.x + w;
    const bool live = k < p.K;                                // the last workgroup of an odd launch: wave 1 only keeps the barriers company
    if (!p.yield_issue) __builtin_amdgcn_s_setprio(3);

    Bin bin;
    bin.status = 2;                                           // (idle wave)
    bin.normS2 = 0;
    bin.sexp = 0;
    bin.rx = bin.ry = 0;
    if (live) front_half<XT>(p, pXB, pXD, pd, k, sC[w], sW[w], bin, lane);
    if (lane == 0) sTrust[w] = (live && bin.status == 0) ? 1 : 0;          // "this bin takes part in the sweeps"
    if (lane == 1 && w < 2) reinterpret_cast<float*>(&sLam2[w][0])[0] = (float)bin.normS2;
    wsync();                                                  // the LDS writes above have landed
    __builtin_amdgcn_s_barrier();
    if (w == 0) {
        // ---- the float pre-solve of both bins at once ----
        const int half = lane >> 5, a = lane & 3, b = (lane >> 2) & 7;
        const CF* const fG = reinterpret_cast<const CF*>(sW[half]);
        const bool take = sTrust[half] != 0;
        CF top[4], bot[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // a bin that is not positive definite (or absent) takes no part: its half rotates an identity
            top[t] = take ? fG[(4 * a + t) * LDF + b] : mk<float>((4 * a + t == b) ? 1.f : 0.f, 0.f);
            bot[t] = take ? fG[(4 * a + t) * LDF + 8 + b] : mk<float>((4 * a + t == 8 + b) ? 1.f : 0.f, 0.f);
        }
        const float nS2 = reinterpret_cast<const float*>(&sLam2[half][0])[0];
        wsync();
        constexpr float kPresolveTol2 = 1e-6f;
        bool fconv;
        float n2t, n2b;
        jacobi16_onesided_x2<LDF>(top, bot, lane, kPresolveTol2, nS2 > 0.f ? nS2 : 1.f, Prec<float>::max_sweeps, fconv, n2t, n2b,
                                  reinterpret_cast<CF*>(sW[half]));
        // trust: converged and eigenvalue span below 1e3, per bin (half of the wave)
        float mn = fminf(n2t, n2b), mx = fmaxf(n2t, n2b);
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) {
            mn = fminf(mn, __shfl_xor(mn, m, 64));
            mx = fmaxf(mx, __shfl_xor(mx, m, 64));
        }
        const bool trust = fconv && (mn >= 1e-3f * mx);                          // NaN counts as not ok; uniform over the half
        wsync();
        // V32 of each bin into its work matrix as float64, columns where the schedule leaves them
        CD* const sV = sW[half];
        const int ct = os_top_end(b), cb = os_bot_end(b);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            sV[(4 * a + t) * LD + ct] = mk<double>((double)top[t].x, (double)top[t].y);
            sV[(4 * a + t) * LD + cb] = mk<double>((double)bot[t].x, (double)bot[t].y);
        }
        if ((lane & 31) == 0) sTrust[half] = (take && trust) ? 1 : 0;
        wsync();
    }
    __builtin_amdgcn_s_barrier();
    if (!live) return;
    int status = bin.status;
    if (status == 0 && sTrust[w] == 0) status = 3;
    if (status == 0) status = back_half(p, pw, plam, k, sC[w], sW[w], sLam[w], sLam2[w], sOrder[w], bin, lane);
    if (status == 1) {
        // not positive definite: zero filters and eigenvalues, status 1 (numpy.linalg.LinAlgError at apvast.py:24)
        if (lane < N) {
            for (int t = 0; t < p.nV; ++t) {
                const size_t oi = ((size_t)k * p.nV + t) * N + lane;
                if (p.out_c128) reinterpret_cast<double2*>(pw)[oi] = make_double2(0.0, 0.0);
                else reinterpret_cast<float2*>(pw)[oi] = make_float2(0.f, 0.f);
            }
            if (plam != nullptr) {
                if (p.out_c128) reinterpret_cast<double*>(plam)[(size_t)k * N + lane] = 0.0;
                else reinterpret_cast<float*>(plam)[(size_t)k * N + lane] = 0.f;
            }
        }
    }
    if (status == 3 && lane == 0) {
        // the one-bin kernel recomputes this bin (its double sweeps): append (zone, bin) to the redo list
        const int slot = atomicAdd(p.redo_count, 1);
        p.redo_list[slot] = k | (z1 ? (1 << 30) : 0);
    }
    if (pstatus != nullptr && lane == 0 && status != 3) pstatus[k] = status;
}

}  // namespace

// hipErrorNotSupported when the launch does not qualify: the one-bin kernel takes it
hipError_t apv_launch_gevd16x2(const GevdParams& p, int compute_dtype, bool fused, hipStream_t s) {
    static const bool off = (getenv("APV_NO_GEVD16X2") != nullptr);        // A/B switch: one bin per wave
    if (off || p.n != 16 || compute_dtype != APV_F64 || !fused || p.reg_mode != APV_REG_ABS || p.reg_bright != 0.0 || p.sweep_tol2 > 0.0 ||
        p.max_sweeps > 0 || p.debug_stop != 0 || p.stamps != nullptr || p.U != nullptr || p.redo_list == nullptr || p.redo_count == nullptr)
        return hipErrorNotSupported;
    if (p.K <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(p.redo_count, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    const dim3 grid((p.K + 1) / 2, p.n_zones > 1 ? 2 : 1);
    if (p.x_c128) hipLaunchKernelGGL((gevd16x2_kernel<double2>), grid, dim3(128), 0, s, p);
    else hipLaunchKernelGGL((gevd16x2_kernel<float2>), grid, dim3(128), 0, s, p);
    return hipGetLastError();
}
