// Device helpers of the order-16 kernel (kernels_gevd16m.hip).
#pragma once
#include <type_traits>
#include "apv_internal.h"

namespace {

constexpr int N = 16;
constexpr int LD = 17;        // row stride of the LDS matrices, in complex elements

template <typename T> struct Cx { T x, y; };
template <typename T> __device__ __forceinline__ Cx<T> mk(T a, T b) { Cx<T> r; r.x = a; r.y = b; return r; }

template <typename T> struct Prec;
template <> struct Prec<double> {
    // a sweep that met off^2/||C||^2 <= tol2 leaves ~tol2^2 behind (quadratic convergence): it is the last one.  1e-10 is
    // enough for spectra within a few decades; on the reference's own room responses (cfg1, cond(R_D) ~ 1e5) it left the
    // filters at 6e-4 of the oracle's, 1e-14 at 5e-8, 1e-18 at 4e-11 (tools/probes/cfg1_square_probe.py)
    static constexpr double sweep_tol2 = 1e-16;
    static constexpr int max_sweeps = 14;
};
template <> struct Prec<float> {
    static constexpr float sweep_tol2 = 1e-8f;
    static constexpr int max_sweeps = 12;
};

// 1/sqrt(x), full precision of T, x > 0 finite
__device__ __forceinline__ double rsq_full(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double t = x * y;
    const double e = __builtin_fma(-t, y, 1.0);
    const double pp = __builtin_fma(0.375, e, 0.5);
    const double ye = y * e;
    // v_rsq_f64 is good to 5e-8 on gfx950 (measured); one third-order step brings it to 1.4e-16
    return __builtin_fma(ye, pp, y);
}
// 1 / x to full precision without the division sequence: v_rcp (5e-8 in double here) and two Newton steps
__device__ __forceinline__ double rcp_full(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = y * __builtin_fma(-x, y, 2.0);
    return y * __builtin_fma(-x, y, 2.0);
}
__device__ __forceinline__ float rcp_full(float x) {
    const float y = __builtin_amdgcn_rcpf(x);
    return y * __builtin_fmaf(-x, y, 2.0f);
}
__device__ __forceinline__ float rsq_full(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    const float t = x * y;
    const float e = __builtin_fmaf(-t, y, 1.0f);
    return __builtin_fmaf(y * e, 0.5f, y);
}

// wave-level ordering point between phases that exchange data through LDS (one wave per workgroup)
__device__ __forceinline__ void wsync() { __syncthreads(); }

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return __shfl(v, 0, 64);
}

// a value that is the same in every lane, moved to scalar registers (the compiler cannot prove uniformity of a shuffle's result)
__device__ __forceinline__ float uniform_scalar(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double uniform_scalar(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

using d4 = __attribute__((ext_vector_type(4))) double;
using f4 = __attribute__((ext_vector_type(4))) float;

// R = X^H X for one [M][16] slab of c64 (XT = float2) or c128 (XT = double2) elements, result written to dst (LDS, row
// stride LD); optional r = X^H d -> sr.  Lane l loads slab element 64 s + l: fully coalesced, and that one register is
// both the A (X^H) and the B (X) operand of the 16x16x4 MFMA.
template <typename T> struct Mfma16;
// a b + c in one rounding, float or double (written a += p q + r s the compiler emits a product, an FMA and an addition)
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <> struct Mfma16<double> {
    using V = d4;
    static __device__ __forceinline__ V mac(double a, double b, V c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    // f64 16x16x4 accumulator: row = (lane>>4) + 4*reg, col = lane&15
    static __device__ __forceinline__ int row(int lane, int t) { return (lane >> 4) + 4 * t; }
};
template <> struct Mfma16<float> {
    using V = f4;
    static __device__ __forceinline__ V mac(float a, float b, V c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    // f32 16x16x4 accumulator: row = 4*(lane>>4) + reg, col = lane&15
    static __device__ __forceinline__ int row(int lane, int t) { return 4 * (lane >> 4) + t; }
};

struct NoStamp { __device__ __forceinline__ void operator()(int) const {} };
// `stamp(i)`: diagnostic hook (stage stamps of the diagnostic kernel instantiation; NoStamp in the product)
// WITH_D: also r = X^H d (a compile-time flag: tested at run time, `dvec != nullptr` put every d load into a basic block of its own
// that ended in `s_waitcnt vmcnt(0)`, i.e. waited for the x loads before it as well)
// GS: element stride of the slab (1: bin-major [K][M][L]; 4: the grouped layout [K/4][M L][4] of the float64 streaming front-end,
// where X points at the bin's first element inside its group and consecutive slab elements are four apart)
template <typename T, typename XT, bool WITH_D, typename ST = NoStamp, int GS = 1>
__device__ __forceinline__ void correlate16(const XT* __restrict__ X, const XT* __restrict__ dvec, int M,
                                            Cx<T>* dst, Cx<T>* sr, int lane, ST stamp = ST(), int stamp_base = 0) {
    using MM = Mfma16<T>;
    typename MM::V re = {0, 0, 0, 0}, im = {0, 0, 0, 0};
    T rx = 0, ry = 0;
    const int msub = lane >> 4;
    XT zero;
    zero.x = 0;
    zero.y = 0;
    // 32 control points (8 k-steps) at a time: all the loads of a chunk are in flight before its first MFMA.  `full` (uniform):
    // M is a multiple of 32, no row needs clamping and no value masking -- 90 selects and 30 address instructions less per bin
    auto chunk = [&](int mc, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        XT xv[8], dv[8];
        const XT* const xrow0 = X + ((size_t)(mc + msub) * N + (lane & 15)) * GS;
        const XT* const drow0 = WITH_D ? dvec + mc + msub : nullptr;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            // UNCONDITIONAL loads (from a clamped row when the chunk is ragged), then a select: written as `ok ? X[...] : zero`
            // every load sat in a branch of its own (the address might be out of bounds), the compiler split it into two dword
            // loads and put an `s_waitcnt vmcnt(0)` behind each -- 48 memory round trips one after the other per bin, a quarter
            // of a wave's life (profiles/r03/stage_stamps_32768.md: 24 k + 12 k cycles "waiting for the slabs")
            const int m = mc + 4 * q + msub;
            const bool ok = FULL || m < M;
            XT xl, dl = zero;
            if constexpr (FULL) {
                // one base address per chunk, the eight rows at constant offsets (512 q bytes: immediate offsets of the loads;
                // indexed per load the compiler spent three 64-bit address instructions on each)
                xl = xrow0[(size_t)(4 * q) * N * GS];
                if constexpr (WITH_D) dl = drow0[4 * q];
            } else {
                const int mcl = ok ? m : M - 1;
                xl = X[((size_t)mcl * N + (lane & 15)) * GS];
                if constexpr (WITH_D) dl = dvec[mcl];
            }
            xv[q].x = ok ? xl.x : zero.x;
            xv[q].y = ok ? xl.y : zero.y;
            dv[q].x = ok ? dl.x : zero.x;
            dv[q].y = ok ? dl.y : zero.y;
        }
        if constexpr (!__is_same(ST, NoStamp)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp(stamp_base);                                   // the chunk's loads have landed
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const T xr = (T)xv[q].x, xi = (T)xv[q].y;
            re = MM::mac(xr, xr, re);
            re = MM::mac(xi, xi, re);
            im = MM::mac(xr, xi, im);                            // P = sum xr (x) xi; Im R = P - P^T, formed below
            if constexpr (WITH_D) {
                rx = fma_t(xi, (T)dv[q].y, fma_t(xr, (T)dv[q].x, rx));        // conj(x) * d, as chained FMAs
                ry = fma_t(-xi, (T)dv[q].x, fma_t(xr, (T)dv[q].y, ry));
            }
        }
    };
    if ((M & 31) == 0) {
        for (int mc = 0; mc < M; mc += 32) chunk(mc, std::true_type{});
    } else {
        for (int mc = 0; mc < M; mc += 32) chunk(mc, std::false_type{});
    }
    // Im R = P - P^T: three MFMAs per k-step instead of four (the f64 matrix pipe is busy half of this kernel's time at the rate
    // the instruction sustains, profiles/r02/mfma_issue_rate.md); the transpose goes through the destination tile
    const int col = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) dst[MM::row(lane, t) * LD + col] = mk<T>(re[t], im[t]);
    stamp(stamp_base + 1);                                   // MFMAs retired (the stores above read the accumulators)
    wsync();
    T pt[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) pt[t] = dst[col * LD + MM::row(lane, t)].y;
    wsync();
#pragma unroll
    for (int t = 0; t < 4; ++t) dst[MM::row(lane, t) * LD + col].y = im[t] - pt[t];
    if constexpr (WITH_D) {
        rx += __shfl_xor(rx, 16, 64); ry += __shfl_xor(ry, 16, 64);
        rx += __shfl_xor(rx, 32, 64); ry += __shfl_xor(ry, 32, 64);
        if (lane < N) sr[lane] = mk<T>(rx, ry);
    }
}



// XOR-schedule for the register-resident Jacobi (JAC == 1).  Pairs of round r are {i, i^r}; the 15 values of r
// are visited grouped by their highest (odd sweeps: lowest) set bit so that between two rounds only the
// "bottom" member of every pair changes slot, by a slot-XOR of 1, 2 or 4.  Entry = {transition bit or -1, delta}.
struct XStep { signed char tbit, delta; };
constexpr XStep XSCHED[2][15] = {
    {{-1, 1}, {-1, 2}, {-1, 1}, {-1, 4}, {-1, 1}, {-1, 2}, {-1, 1}, {-1, 4}, {2, 1}, {-1, 2}, {-1, 1}, {-1, 2}, {1, 1}, {-1, 1}, {0, 0}},
    {{-1, 1}, {-1, 2}, {-1, 1}, {-1, 4}, {-1, 1}, {-1, 2}, {-1, 1}, {-1, 4}, {0, 2}, {-1, 4}, {-1, 2}, {-1, 4}, {1, 4}, {-1, 4}, {2, 0}}};
// the same schedule as nibble strings: nibble r = delta of round r / (transition bit + 1) of round r
constexpr unsigned long long xs_pack(int sweep, bool tb) {
    unsigned long long v = 0;
    for (int r = 0; r < 15; ++r)
        v |= (unsigned long long)(tb ? (XSCHED[sweep][r].tbit + 1) : XSCHED[sweep][r].delta) << (4 * r);
    return v;
}
constexpr unsigned long long XS_DELTA0 = xs_pack(0, false), XS_DELTA1 = xs_pack(1, false);
constexpr unsigned long long XS_TBIT0 = xs_pack(0, true), XS_TBIT1 = xs_pack(1, true);

template <typename T> __device__ __forceinline__ Cx<T> cshfl(Cx<T> v, int src) {
    return mk<T>(__shfl(v.x, src, 64), __shfl(v.y, src, 64));
}
// members of a (top, bottom) pair trade places across lanes `peer`: the lane whose slot bit is set gives its
// top and keeps its bottom, the other gives its bottom and keeps its top
template <typename T> __device__ __forceinline__ void xchg(Cx<T>& top, Cx<T>& bot, bool bit, int peer) {
    const Cx<T> send = bit ? top : bot;
    const Cx<T> recv = cshfl(send, peer);
    if (bit) top = recv; else bot = recv;
}

// ---- cross-lane moves by XOR of the lane id, on the VALU (DPP) where the ISA allows it ----
template <int D> __device__ __forceinline__ int dpp_xor_lo(int v);       // lane ^ D, D in {1,2,4} (within 8 lanes)
// full permutations: no `old` operand, so no copy in front of the DPP move
template <> __device__ __forceinline__ int dpp_xor_lo<1>(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, false); }  // quad_perm [1,0,3,2]
template <> __device__ __forceinline__ int dpp_xor_lo<2>(int v) { return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, false); }  // quad_perm [2,3,0,1]
// lane ^ 4 has no single DPP pattern (it takes row_shl:4 + row_shr:4 under bank masks plus copies); ds_swizzle in
// bit-mask mode (and 0x1f, or 0, xor 4) is one crossbar operation and keeps the VALU, the bound unit, free
template <> __device__ __forceinline__ int dpp_xor_lo<4>(int v) { return __builtin_amdgcn_ds_swizzle(v, 0x101F); }
__device__ __forceinline__ int dpp_xor8(int v) { return __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, false); }  // row_ror:8

template <int D> __device__ __forceinline__ float xcol(float v) { return __int_as_float(dpp_xor_lo<D>(__float_as_int(v))); }
template <int D> __device__ __forceinline__ double xcol(double v) {
    return __hiloint2double(dpp_xor_lo<D>(__double2hiint(v)), dpp_xor_lo<D>(__double2loint(v)));
}
template <int D> __device__ __forceinline__ float xrow(float v, int lane) {
    if constexpr (D == 1) return __int_as_float(dpp_xor8(__float_as_int(v)));
    else return __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ (8 * D)) << 2, __float_as_int(v)));
}
template <int D> __device__ __forceinline__ double xrow(double v, int lane) {
    if constexpr (D == 1) return __hiloint2double(dpp_xor8(__double2hiint(v)), dpp_xor8(__double2loint(v)));
    else {
        const int addr = (lane ^ (8 * D)) << 2;
        return __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, __double2hiint(v)),
                                __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)));
    }
}
// ---- top/bottom exchanges of the schedule's transitions as single-instruction lane swaps ----
// Rows (lane bits 3, 4, 5): the lanes whose bit is set give their top and take the partner's bottom.
//   bit 3: row_ror:8 under bank masks; bit 4: v_permlane16_swap (odd 16-lane rows of `top` <-> even rows of `bot`);
//   bit 5: v_permlane32_swap (upper half of `top` <-> lower half of `bot`)
template <int TB> __device__ __forceinline__ void xswap_row(int& top, int& bot) {
    if constexpr (TB == 0) {
        const int nt = __builtin_amdgcn_update_dpp(top, bot, 0x128, 0xf, 0xc, false);
        bot = __builtin_amdgcn_update_dpp(bot, top, 0x128, 0xf, 0x3, false);
        top = nt;
    } else if constexpr (TB == 1) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)top, (unsigned)bot, false, false);
        top = (int)r[0];
        bot = (int)r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)top, (unsigned)bot, false, false);
        top = (int)r[0];
        bot = (int)r[1];
    }
}
// Columns, lane bit 2: row_shr:4 / row_shl:4 under bank masks (bits 0 and 1 have no masked form: generic xchg)
__device__ __forceinline__ void xswap_col4(int& top, int& bot) {
    const int nt = __builtin_amdgcn_update_dpp(top, bot, 0x114, 0xf, 0xa, false);
    bot = __builtin_amdgcn_update_dpp(bot, top, 0x104, 0xf, 0x5, false);
    top = nt;
}
template <int TB> __device__ __forceinline__ void xswap_row(float& t, float& b) {
    int ti = __float_as_int(t), bi = __float_as_int(b);
    xswap_row<TB>(ti, bi);
    t = __int_as_float(ti);
    b = __int_as_float(bi);
}
template <int TB> __device__ __forceinline__ void xswap_row(double& t, double& b) {
    int tl = __double2loint(t), th = __double2hiint(t), bl = __double2loint(b), bh = __double2hiint(b);
    xswap_row<TB>(tl, bl);
    xswap_row<TB>(th, bh);
    t = __hiloint2double(th, tl);
    b = __hiloint2double(bh, bl);
}
__device__ __forceinline__ void xswap_col4(float& t, float& b) {
    int ti = __float_as_int(t), bi = __float_as_int(b);
    xswap_col4(ti, bi);
    t = __int_as_float(ti);
    b = __int_as_float(bi);
}
__device__ __forceinline__ void xswap_col4(double& t, double& b) {
    int tl = __double2loint(t), th = __double2hiint(t), bl = __double2loint(b), bh = __double2hiint(b);
    xswap_col4(tl, bl);
    xswap_col4(th, bh);
    t = __hiloint2double(th, tl);
    b = __hiloint2double(bh, bl);
}
template <int TB, typename T> __device__ __forceinline__ void cxswap_row(Cx<T>& t, Cx<T>& b) {
    xswap_row<TB>(t.x, b.x);
    xswap_row<TB>(t.y, b.y);
}
template <typename T> __device__ __forceinline__ void cxswap_col4(Cx<T>& t, Cx<T>& b) {
    xswap_col4(t.x, b.x);
    xswap_col4(t.y, b.y);
}

template <int D, typename T> __device__ __forceinline__ Cx<T> cxcol(Cx<T> v) { return mk<T>(xcol<D>(v.x), xcol<D>(v.y)); }
template <int D, typename T> __device__ __forceinline__ Cx<T> cxrow(Cx<T> v, int lane) { return mk<T>(xrow<D>(v.x, lane), xrow<D>(v.y, lane)); }

// bottoms of every pair move by slot-XOR D: columns (lane bits 0-2) and rows (lane bits 3-5)
template <int D, typename T>
__device__ __forceinline__ void move_bottoms(Cx<T>& tb, Cx<T>& bt, Cx<T>& bb, Cx<T>& v0b, Cx<T>& v1b, int lane) {
    tb = cxcol<D>(tb);
    v0b = cxcol<D>(v0b);
    v1b = cxcol<D>(v1b);
    bb = cxcol<D>(bb);
    bt = cxrow<D>(bt, lane);
    bb = cxrow<D>(bb, lane);
}

// ---- applying a rotation J = [[c, s], [-conj(s), c]] to a pair of complex numbers ----
//   columns  [p, q] J      : p' = c p - conj(s) q,  q' = c q + s p
//   rows     J^H [p; q]    : p' = c p - s q,        q' = c q + conj(s) p
// double: scalar FMAs.  float: the (re, im) pair is one 64-bit operand of v_pk_mul_f32 / v_pk_fma_f32 (the op_sel and
// neg modifiers supply the swaps and sign flips of a complex product), which halves the instruction count.
template <typename T>
__device__ __forceinline__ void rot_cols(T c, Cx<T> s, Cx<T> p, Cx<T> q, Cx<T>& po, Cx<T>& qo) {
    po.x = c * p.x - (s.x * q.x + s.y * q.y);
    po.y = c * p.y - (s.x * q.y - s.y * q.x);
    qo.x = c * q.x + (s.x * p.x - s.y * p.y);
    qo.y = c * q.y + (s.x * p.y + s.y * p.x);
}
template <typename T>
__device__ __forceinline__ void rot_rows(T c, Cx<T> s, Cx<T> p, Cx<T> q, Cx<T>& po, Cx<T>& qo) {
    po.x = c * p.x - (s.x * q.x - s.y * q.y);
    po.y = c * p.y - (s.x * q.y + s.y * q.x);
    qo.x = c * q.x + (s.x * p.x + s.y * p.y);
    qo.y = c * q.y + (s.x * p.y - s.y * p.x);
}
using f2v = __attribute__((ext_vector_type(2))) float;
using f4v = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f2v pk_cmul(f2v s, f2v z) {          // s z
    const f2v t = {-s.y, s.y};
    return __builtin_elementwise_fma((f2v){s.x, s.x}, z, t * (f2v){z.y, z.x});
}
__device__ __forceinline__ f2v pk_cmulc(f2v s, f2v z) {         // conj(s) z
    const f2v t = {s.y, -s.y};
    return __builtin_elementwise_fma((f2v){s.x, s.x}, z, t * (f2v){z.y, z.x});
}
template <>
__device__ __forceinline__ void rot_cols<float>(float c, Cx<float> s, Cx<float> p, Cx<float> q, Cx<float>& po, Cx<float>& qo) {
    const f2v cc = {c, c}, sv = {s.x, s.y}, pv = {p.x, p.y}, qv = {q.x, q.y};
    const f2v a = cc * pv - pk_cmulc(sv, qv), b = cc * qv + pk_cmul(sv, pv);
    po = mk<float>(a.x, a.y);
    qo = mk<float>(b.x, b.y);
}
template <>
__device__ __forceinline__ void rot_rows<float>(float c, Cx<float> s, Cx<float> p, Cx<float> q, Cx<float>& po, Cx<float>& qo) {
    const f2v cc = {c, c}, sv = {s.x, s.y}, pv = {p.x, p.y}, qv = {q.x, q.y};
    const f2v a = cc * pv - pk_cmul(sv, qv), b = cc * qv + pk_cmulc(sv, pv);
    po = mk<float>(a.x, a.y);
    qo = mk<float>(b.x, b.y);
}

// Jacobi rotation J = [[c, s], [-conj(s), c]] for the Hermitian 2x2 [[alpha, beta], [conj(beta), gamma]].
// Any complex t gives an exactly unitary J once c = 1/sqrt(1+|t|^2), s = t c are formed in T, so the
// angle t = sign(tau) e^{i arg beta} / (|tau| + sqrt(1+tau^2)) is evaluated in float (relative 1e-7: the pair's
// off-diagonal drops by that factor instead of to zero, which the next sweep finishes).  Inputs are
// pre-scaled to ||C||_F ~ 1, so float range is not an issue; |beta|^2 < 1e-30 is skipped.
template <typename T>
__device__ __forceinline__ void rotation(T alpha, T gamma, T bx, T by, T& c, T& sx, T& sy) {
    const float fbx = (float)bx, fby = (float)by, fd = (float)(gamma - alpha);
    const float b2 = fbx * fbx + fby * fby;
    if constexpr (sizeof(T) == 4) {
        // float (the order-64 kernel's pair solves, whose dependent chain is what a block round waits for): the closed form of
        // the one-sided rounds -- with zeta = (gamma - alpha)/2, h = sqrt(zeta^2 + |beta|^2), D = |zeta| + h: c = D r,
        // s = sign(zeta) beta r, r = 1/sqrt(2 h D).  Two transcendental instructions and no branch instead of four and one.
        const float zeta = 0.5f * fd;
        const float x = fmaxf(__builtin_fmaf(zeta, zeta, b2), 1e-36f);
        const float hh = x * __builtin_amdgcn_rsqf(x);
        const float D = fmaxf(fabsf(zeta), 1e-18f) + hh;
        const float rr = __builtin_amdgcn_rsqf((hh + hh) * D);
        const float rs = copysignf(rr, zeta);
        c = D * rr;
        sx = fbx * rs;
        sy = fby * rs;
        return;
    }
    float tx = 0.f, ty = 0.f;
    if (b2 > 1e-30f) {
        const float iab = __builtin_amdgcn_rsqf(b2);
        const float tau = fd * 0.5f * iab;
        const float rho = __builtin_amdgcn_sqrtf(__builtin_fmaf(tau, tau, 1.0f));
        const float t = copysignf(__builtin_amdgcn_rcpf(fabsf(tau) + rho), tau) * iab;
        tx = fbx * t;
        ty = fby * t;
    }
    const T dx = (T)tx, dy = (T)ty;
    c = rsq_full((T)1 + dx * dx + dy * dy);
    sx = dx * c;
    sy = dy * c;
}

// The register-resident cyclic Jacobi of stage 3 on the 2 x 2 blocks (tt, tb; bt, bb) of this lane and its two rows
// of V, in precision TT.  Runs sweeps until one of them meets sum |pivot|^2 <= tol2 normS2 (that sweep is the last) or
// max_sweeps is reached; returns the number of sweeps done (its parity says which slot layout the blocks are left in).
// n_rounds < 15 cuts a sweep short: the first 8 rounds of an even sweep are exactly the 64 pairs between the index halves
// {0..7} and {8..15} (slot s pairs s with 8 + (s ^ x), x running through a Gray code) and bring the slots back to where
// they started, which is what the block Jacobi of the order-64 kernel needs for a pair of blocks.
template <typename TT>
__device__ __forceinline__ int jacobi16_sweeps(Cx<TT>& tt_, Cx<TT>& tb_, Cx<TT>& bt_, Cx<TT>& bb_, Cx<TT>& v0t_, Cx<TT>& v0b_,
                                               Cx<TT>& v1t_, Cx<TT>& v1b_, TT (*srot)[4], int lane, TT tol2, TT normS2,
                                               int max_sweeps, bool& converged_, int n_rounds = 15) {
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
        for (int r = 0; r < n_rounds; ++r) {
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


// ONE sweep of schedule 0 with its first NR rounds, the schedule a compile-time constant and the rounds unrolled (every round
// knows its move and its re-deal: no branch, no nibble arithmetic).  NR = 15: a full sweep, the blocks are left in the
// (2s, 2s+1) layout; NR = 8: the 64 pairs between the index halves, slots back where they started.  This is what the block
// Jacobi of the order-64 kernel calls for its pair problems.
template <typename TT, int NR>
__device__ __forceinline__ void jacobi16_sweep0(Cx<TT>& tt_, Cx<TT>& tb_, Cx<TT>& bt_, Cx<TT>& bb_, Cx<TT>& v0t_, Cx<TT>& v0b_,
                                                Cx<TT>& v1t_, Cx<TT>& v1b_, TT (*srot)[4], int lane) {
    using CC = Cx<TT>;
    const int a = lane >> 3, b = lane & 7;
    const bool diag = (a == b);
    CC tt = tt_, tb = tb_, bt = bt_, bb = bb_, v0t = v0t_, v0b = v0b_, v1t = v1t_, v1b = v1b_;
    TT off = 0;
    {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int delta = (int)((XS_DELTA0 >> (4 * r)) & 15), tbit = (int)((XS_TBIT0 >> (4 * r)) & 15) - 1;
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
    }
    (void)off;
    tt_ = tt; tb_ = tb; bt_ = bt; bb_ = bb; v0t_ = v0t; v0b_ = v0b; v1t_ = v1t; v1b_ = v1b;
}

// ---- float32 ONE-SIDED Jacobi (Hestenes) for the pre-solve of the float64 order-16 kernel -------------------------------
// With G G^H = C, column rotations G <- G J leave G G^H alone and end with orthogonal columns G J = U Sigma, so the
// normalised columns ARE the eigenvectors of C: no accumulation of V, no two-sided update of C.  A round rotates one
// 16 x 16 array once (12 packed operations per lane) where the two-sided form rotates three (36), the pivots
// g_p^H g_q come from an all-reduce over the eight lanes that share a column slot (so every lane forms its rotation itself:
// no broadcast), and only the columns move between lanes.  The schedule, the slot layout and the moves are those of
// jacobi16_sweeps; lane a + 8 b holds rows 2a, 2a+1 of the columns in slot b (top, bottom).

// Lane layout of the one-sided solve: lane = a + 8 b, a = row pair (rows 2a, 2a+1), b = column slot -- the transpose of the
// two-sided layout, chosen so that the sum over the eight lanes that share a slot (a = lane bits 0-2) is three DPP adds on
// the VALU (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror: after the first two every lane of a quad holds the quad's
// sum, and the mirror pairs quad 0 with quad 1) instead of a DPP add and two round trips over the LDS crossbar: a Jacobi round
// has one such round trip left (the column moves), where the serial chain of the round used to have three.
// (update_dpp with a zero `old` and bound_ctrl is the form the DPP combiner folds into the add itself, v_add_f32_dpp.)
__device__ __forceinline__ float colsum8(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
    return v;
}
__device__ __forceinline__ void xchg_f(float& top, float& bot, bool bit, int peer) {
    const float send = bit ? top : bot;
    const float recv = __shfl(send, peer, 64);
    if (bit) top = recv; else bot = recv;
}

// float32 Cholesky factor of (2^sexp C + delta I) from the matrix in sA, left in fG [16][LDF].  The elimination runs from
// the LAST index to the first (the indices are reversed on the way in and on the way out), i.e. fG is the UPPER factor U with
// U U^H = C: on the whitened matrices of this path the one-sided sweeps that follow need 0.7 sweeps fewer from it than from
// the lower factor (5.5 -> 4.8 on the bench workload; a diagonally pivoted factor would save 0.9 and cost a wave-wide argmax
// per step).  Lane (i = lane >> 2, jq = lane & 3) owns row i, columns jq + 4t; one column goes through LDS per step.
// A pivot that is not positive is clamped: the factor only seeds the pre-solve, whose result the float64 refinement
// certifies against the exact C.
__device__ __forceinline__ float scale_to_f32(double v, int e) { return (float)ldexp(v, e); }
__device__ __forceinline__ float scale_to_f32(float v, int e) { return ldexpf(v, e); }
template <typename TS, int LDA, int LDF>
__device__ __forceinline__ void chol16_f32(const Cx<TS>* sA, int sexp, float delta, Cx<float>* fG, Cx<float> (*fcol)[16],
                                           int lane) {
    const int i = lane >> 2, jq = lane & 3;
    // Symmetric permutation first: the indices in ASCENDING order of the diagonal, so that the reversed elimination below takes
    // the largest diagonal element first (round 3; NumPy model of the sweeps, tools/probes/onesided_schedule_model.py: 4.58 ->
    // 4.35 sweeps on the bench workload; pivoting on the CURRENT diagonal at every step gives 4.12 but costs more than it
    // saves: a wave-wide argmax per step and no statically known spent column groups).  Row i counts the diagonal elements
    // below its own, a quarter per lane; G = P U keeps its rows at their own indices, so nothing downstream sees the order.
    int* const sperm = reinterpret_cast<int*>(&fcol[1][0]);           // 16 ints; the staging buffer is first written in step 1
    {
        const TS di = sA[i * LDA + i].x;
        int below = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = jq + 4 * t;
            const TS dj = sA[j * LDA + j].x;
            below += (dj < di || (dj == di && j < i)) ? 1 : 0;
        }
        below += __builtin_amdgcn_update_dpp(0, below, 0xB1, 0xf, 0xf, true);       // over the four lanes of the row
        below += __builtin_amdgcn_update_dpp(0, below, 0x4E, 0xf, 0xf, true);
        if (jq == 0) sperm[below] = i;
    }
    wsync();
    // (& 15: a NaN on the diagonal leaves slots of sperm unwritten -- the result is NaN either way, the indices stay inside the matrix)
    const int pi = sperm[15 - i] & 15;                                // the index that sits at (reversed) position i
    f2v brow[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = jq + 4 * t;
        const Cx<TS> v = sA[pi * LDA + (sperm[15 - j] & 15)];             // the matrix sorted and with its indices reversed: see fG below
        brow[t] = (f2v){scale_to_f32(v.x, sexp), scale_to_f32(v.y, sexp)};
        if (j == i) brow[t] = (f2v){brow[t].x + delta, 0.f};
    }
    wsync();                                                           // sperm is read: the staging buffer may be written
    // The outer-product update runs on the WHOLE Hermitian matrix, without the triangle tests: rows and columns already
    // eliminated only cancel to rounding level and are never read again, and a wave-wide unconditional update is cheaper than
    // its predicates.  The column goes through LDS permuted (element j = jq + 4t at jq*4 + t) so that the four partners of a
    // lane are two 16-byte reads.
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const int buf = kk & 1;
        if (jq == (kk & 3)) fcol[buf][(i & 3) * 4 + (i >> 2)] = mk<float>(brow[kk >> 2].x, brow[kk >> 2].y);
        wsync();
        const float dkk = fmaxf(fcol[buf][(kk & 3) * 4 + (kk >> 2)].x, 1e-12f);
        const float inv = __builtin_amdgcn_rsqf(dkk), inv2 = inv * inv;
        const Cx<float> lic = fcol[buf][(i & 3) * 4 + (i >> 2)];
        if (jq == (kk & 3)) {
            Cx<float> g = mk<float>(lic.x * inv, lic.y * inv);
            if (i == kk) g = mk<float>(dkk * inv, 0.f);
            if (i < kk) g = mk<float>(0.f, 0.f);
            fG[pi * LDF + (15 - kk)] = g;
        }
        const f2v li2 = {lic.x * inv2, lic.y * inv2};
        const f4v* const part = reinterpret_cast<const f4v*>(&fcol[buf][jq * 4]);
        // column groups entirely at or before the pivot (4t + 3 <= kk, known to the unrolled loop) are never read again
        if (kk < 7) {
            const f4v l01 = part[0];
            if (kk < 3) brow[0] -= pk_cmulc((f2v){l01.x, l01.y}, li2);       // B[i][j] -= B[i][kk] conj(B[j][kk]) / d
            brow[1] -= pk_cmulc((f2v){l01.z, l01.w}, li2);
        }
        if (kk < 15) {
            const f4v l23 = part[1];
            if (kk < 11) brow[2] -= pk_cmulc((f2v){l23.x, l23.y}, li2);
            brow[3] -= pk_cmulc((f2v){l23.z, l23.w}, li2);
        }
    }
    wsync();
}

// Sweeps until one of them meets sum |g_p^H g_q|^2 <= tol2 normS2 (that sweep is the last) or max_sweeps is reached; the
// columns come back normalised.  Returns the number of sweeps.
// The stop criterion is absolute: columns of small norm (eigenvalues below ~1e-3 of the largest; the null space of a
// rank-deficient C sits at the shift) weigh nothing in it and may be left askew.  The squared column norms -- the eigenvalues
// of G G^H -- come back with the columns; the caller looks at them and does not use such a result (gevd16m: `trust`).
// The one-sided solve runs ONE schedule, the same pairs in the same order in every sweep, r essentially descending (round r
// pairs index i with i ^ r).  Cyclic Jacobi converges faster when a sweep repeats the previous one than when the pairing is
// re-dealt from sweep to sweep, as the two alternating schedules of jacobi16_sweeps do (NumPy model, tools/probes/
// onesided_schedule_model.py: 5.0 sweeps alternating, 4.7 the same slot schedule with the columns put back where they started,
// 4.5-4.6 orders of this kind; this one is r = 15..8, 7, 5, 6, 4, 2, 3, 1, picked among the 48 that need three re-deals).
// The moves are crossbar permutes here, so a round may shift the bottoms by any slot-XOR (not only the single bits a DPP
// move reaches).  Entry r of the nibble strings: delta / (re-deal bit + 1).
constexpr unsigned long long os_pack(const int (&v)[15]) {
    unsigned long long x = 0;
    for (int r = 0; r < 15; ++r) x |= (unsigned long long)v[r] << (4 * r);
    return x;
}
constexpr int OS_DELTA_V[15] = {7, 1, 3, 1, 7, 1, 3, 1, 3, 2, 3, 2, 0, 1, 0};
constexpr int OS_TBIT_V[15] = {0, 0, 0, 0, 0, 0, 0, 0, 3, 0, 0, 0, 2, 0, 1};
constexpr unsigned long long OS_DELTA = os_pack(OS_DELTA_V), OS_TBIT = os_pack(OS_TBIT_V);
// where the columns are after a sweep that started with slot b holding columns (b, 8 + b): nibble b = column in slot b
constexpr unsigned OS_TOP_END = 0xFCB87430u, OS_BOT_END = 0xEDA96521u;     // tops 0,3,4,7,8,11,12,15; bottoms 1,2,5,6,9,10,13,14
__device__ __forceinline__ int os_top_end(int b) { return (OS_TOP_END >> (4 * b)) & 15; }
__device__ __forceinline__ int os_bot_end(int b) { return (OS_BOT_END >> (4 * b)) & 15; }

// `stage` [16][LDF]: LDS staging through which the columns go back to their starting slots between two sweeps.  After the
// last sweep they are left where the schedule ends (os_top_end / os_bot_end).
template <int LDF>
__device__ __forceinline__ int jacobi16_onesided(Cx<float>& g0t_, Cx<float>& g0b_, Cx<float>& g1t_, Cx<float>& g1b_, int lane,
                                                 float tol2, float normS2, int max_sweeps, bool& converged_, float& n2t_, float& n2b_,
                                                 Cx<float>* stage) {
    using CC = Cx<float>;
    const int a = lane & 7, b = lane >> 3, lane4 = lane << 2;
    int sweeps_done = 0;
    bool converged = false;
    CC g0t = g0t_, g0b = g0b_, g1t = g1t_, g1b = g1b_;
    auto norm2 = [&](CC x, CC y) { return colsum8(x.x * x.x + x.y * x.y + y.x * y.x + y.y * y.y); };
    for (int sweep = 0; sweep < max_sweeps && !converged; ++sweep) {
        if (sweep > 0) {
            // columns back to their starting slots: the same pairs meet in the same order in every sweep
            wsync();
            stage[(2 * a) * LDF + os_top_end(b)] = g0t; stage[(2 * a) * LDF + os_bot_end(b)] = g0b;
            stage[(2 * a + 1) * LDF + os_top_end(b)] = g1t; stage[(2 * a + 1) * LDF + os_bot_end(b)] = g1b;
            wsync();
            g0t = stage[(2 * a) * LDF + b]; g0b = stage[(2 * a) * LDF + 8 + b];
            g1t = stage[(2 * a + 1) * LDF + b]; g1b = stage[(2 * a + 1) * LDF + 8 + b];
        }
        float off = 0.f;
        // squared column norms: formed afresh every sweep, carried through the rotations inside it
        float nt = norm2(g0t, g1t), nb = norm2(g0b, g1b);
        // the schedule is a compile-time constant: unrolled, every round knows its move and its re-deal (no branch, no nibble
        // arithmetic; 896 -> 790 VALU and 280 -> 140 SALU instructions per sweep)
#pragma unroll
        for (int r = 0; r < 15; ++r) {
            const int delta = (int)((OS_DELTA >> (4 * r)) & 15), tbit = (int)((OS_TBIT >> (4 * r)) & 15) - 1;
            // re-deal of tops and bottoms between slots b and b ^ (1 << tbit), three times per sweep: the slot is lane bits 3-5,
            // for which the exchange is a masked row_ror:8 pair, one v_permlane16_swap or one v_permlane32_swap per register
            if (tbit == 0) { cxswap_row<0>(g0t, g0b); cxswap_row<0>(g1t, g1b); xswap_row<0>(nt, nb); }
            else if (tbit == 1) { cxswap_row<1>(g0t, g0b); cxswap_row<1>(g1t, g1b); xswap_row<1>(nt, nb); }
            else if (tbit == 2) { cxswap_row<2>(g0t, g0b); cxswap_row<2>(g1t, g1b); xswap_row<2>(nt, nb); }
            if (delta == 1) {
                // slot ^ 1 = lane ^ 8: row_ror:8 on the VALU, no trip over the crossbar (the schedule is a compile-time constant and
                // the round loop unrolls, so this test costs nothing)
                auto mv = [&](float v) { return __int_as_float(dpp_xor8(__float_as_int(v))); };
                g0b = mk<float>(mv(g0b.x), mv(g0b.y));
                g1b = mk<float>(mv(g1b.x), mv(g1b.y));
                nb = mv(nb);
            } else if (delta != 0) {
                // the bottoms move by slot-XOR delta: five crossbar permutes under one address
                const int addr = lane4 ^ (delta << 5);                 // lane ^ (8 delta)
                auto mv = [&](float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); };
                g0b = mk<float>(mv(g0b.x), mv(g0b.y));
                g1b = mk<float>(mv(g1b.x), mv(g1b.y));
                nb = mv(nb);
            }
            // pivot beta = g_top^H g_bottom over the 16 rows
            // conj(t) b summed over the lane's two rows as A + (B.x, -B.y), A = sum t.x (b.x, b.y), B = sum t.y (b.y, b.x): one
            // sign at the end (written per product the compiler builds a (t.y, -t.y) vector for each row, two extra instructions)
            const f2v pa = __builtin_elementwise_fma((f2v){g1t.x, g1t.x}, (f2v){g1b.x, g1b.y}, (f2v){g0t.x, g0t.x} * (f2v){g0b.x, g0b.y});
            const f2v pb = __builtin_elementwise_fma((f2v){g1t.y, g1t.y}, (f2v){g1b.y, g1b.x}, (f2v){g0t.y, g0t.y} * (f2v){g0b.y, g0b.x});
            const f2v part = {pa.x + pb.x, pa.y - pb.y};
            // The two components are reduced as two scalar chains, interleaved stage by stage: paired into v_pk_add_f32 by the
            // vectoriser they need a v_mov_b32_dpp per component and stage plus the DPP wait states after every add; alone each
            // stage is ONE v_add_f32_dpp, and the other chain's add fills one of the two wait states.  (update_dpp with a zero
            // `old` and bound_ctrl is the form the DPP combiner folds into the add; the empty asm statements keep the
            // vectoriser from pairing the chains and pin the order.)
            float bx = part.x, by = part.y;
#define APV_DPP_ADD(v, ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, true))
            APV_DPP_ADD(bx, 0xB1); asm volatile("" : "+v"(bx)); APV_DPP_ADD(by, 0xB1); asm volatile("" : "+v"(by));
            APV_DPP_ADD(bx, 0x4E); asm volatile("" : "+v"(bx)); APV_DPP_ADD(by, 0x4E); asm volatile("" : "+v"(by));
            APV_DPP_ADD(bx, 0x141); asm volatile("" : "+v"(bx)); APV_DPP_ADD(by, 0x141);
#undef APV_DPP_ADD
            const float b2 = bx * bx + by * by;
            off += b2;
            // rotation for [[nt, beta], [conj(beta), nb]]: with zeta = (nb - nt)/2 and D = |zeta| + sqrt(zeta^2 + |beta|^2),
            // t = sign(zeta) beta / D (|t| <= 1; no division by |beta|), c = 1/sqrt(1 + |t|^2), s = t c; the diagonal moves by
            // Re(conj(t) beta) = sign(zeta) |beta|^2 / D.  Three transcendental instructions, no branch: zeta = beta = 0 gives t = 0.
            // (round 3: TWO transcendental instructions.  With h = sqrt(zeta^2 + |beta|^2): D^2 + |beta|^2 = 2 h D, so
            // c = 1/sqrt(1 + |t|^2) = D r and s = sign(zeta) beta r with r = 1/sqrt(2 h D), and 1/D = 2 h r^2 gives the diagonal
            // shift without a reciprocal.  The floor on |zeta| makes zeta = beta = 0 come out as c = 1, s = 0.)
            const float zeta = 0.5f * (nb - nt);
            const float x = fmaxf(__builtin_fmaf(zeta, zeta, b2), 1e-36f);
            const float hh = x * __builtin_amdgcn_rsqf(x);
            const float D = fmaxf(fabsf(zeta), 1e-18f) + hh;
            const float h2 = hh + hh;
            const float rr = __builtin_amdgcn_rsqf(h2 * D);
            const float c = D * rr;
            const float rs = copysignf(rr, zeta);
            const CC s = mk<float>(bx * rs, by * rs);
            const float shift = b2 * h2 * rr * rs;
            nt -= shift;
            nb += shift;
            CC w0p, w0q, w1p, w1q;
            rot_cols<float>(c, s, g0t, g0b, w0p, w0q);
            rot_cols<float>(c, s, g1t, g1b, w1p, w1q);
            g0t = w0p; g0b = w0q; g1t = w1p; g1b = w1q;
        }
        ++sweeps_done;
        // every pair was counted by the eight lanes of its slot
        const float tot = wave_sum(off) * 0.125f;
        if (tot <= tol2 * normS2) converged = true;
    }
    // squared column norms = eigenvalues of G G^H
    n2t_ = norm2(g0t, g1t);
    n2b_ = norm2(g0b, g1b);
    const float it = __builtin_amdgcn_rsqf(fmaxf(n2t_, 1e-30f)), ib = __builtin_amdgcn_rsqf(fmaxf(n2b_, 1e-30f));
    g0t_ = mk<float>(g0t.x * it, g0t.y * it); g1t_ = mk<float>(g1t.x * it, g1t.y * it);
    g0b_ = mk<float>(g0b.x * ib, g0b.y * ib); g1b_ = mk<float>(g1b.x * ib, g1b.y * ib);
    converged_ = converged;
    return sweeps_done;
}

// ---- complex 16 x 16 x 16 products on the matrix cores (shared by the order-16 kernels) ----------------------------------------
// complex 16x16x16 product on the matrix cores.  fa(i, k) / fb(k, j) fetch operand elements; out[t] is the
// element (mfma_row<T>(lane, t), lane & 15).
template <typename T> __device__ __forceinline__ int mfma_row(int lane, int t);
template <> __device__ __forceinline__ int mfma_row<double>(int lane, int t) { return (lane >> 4) + 4 * t; }
template <> __device__ __forceinline__ int mfma_row<float>(int lane, int t) { return 4 * (lane >> 4) + t; }

// float64: three real products per k-step instead of four (Karatsuba), in three passes over the k-steps so that two accumulators
// suffice: P1 = sum ar br, P2 = sum ai bi; re = P1 - P2; the third pass accumulates sum (ar + ai)(br + bi) onto -(P1 + P2), which
// is the imaginary part.  The f64 matrix pipe is busy ~45 % of this kernel's time at the rate the instruction sustains
// (profiles/r02/mfma_issue_rate.md), barely overlapped with the VALU: 12 MFMAs and ~30 VALU instructions beat 16 MFMAs.
template <typename FA, typename FB>
__device__ __forceinline__ void cmm16(FA fa, FB fb, int lane, Cx<double> out[4]) {
    d4 p1 = {0, 0, 0, 0}, p2 = {0, 0, 0, 0};
    const int rc = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const Cx<double> a = fa(rc, 4 * s + kq), b = fb(4 * s + kq, rc);
        p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, p1, 0, 0, 0);
        p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.y, p2, 0, 0, 0);
    }
    d4 im;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const double u = p1[t], v = p2[t];
        p1[t] = u - v;                                           // re
        im[t] = -u - v;                                          // (two sign modifiers on one add; as -(u + v) it is an add and four
                                                                 //  sign flips.  Which of the two forms spills more has changed with
                                                                 //  the code around it: check WRITE_SIZE after touching this kernel)
    }
    // (the operands are fetched again -- the compiler barrier keeps it from holding the eight complex numbers of the first two
    // passes in registers, which the kernel does not have)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const Cx<double> a = fa(rc, 4 * s + kq), b = fb(4 * s + kq, rc);
        im = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x + a.y, b.x + b.y, im, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) out[t] = mk<double>(p1[t], im[t]);
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

// The same product with operand lambdas that also receive the k-step s: the accumulator layout of a product X,
// out[t] = X[(lane >> 4) + 4 t][lane & 15], is at once the B operand X of the next product (fb = out[s]) and the A operand
// X^T (fa = out[s]), so a result can feed a product without a round trip through LDS.
template <typename FA, typename FB>
__device__ __forceinline__ void cmm16x(FA fa, FB fb, int lane, Cx<double> out[4]) {
    d4 re = {0, 0, 0, 0}, im = {0, 0, 0, 0};
    const int rc = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const Cx<double> a = fa(s, rc, 4 * s + kq), b = fb(s, 4 * s + kq, rc);
        re = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, re, 0, 0, 0);
        re = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.y, b.y, re, 0, 0, 0);
        im = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.y, im, 0, 0, 0);
        im = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.x, im, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) out[t] = mk<double>(re[t], im[t]);
}


}  // namespace
