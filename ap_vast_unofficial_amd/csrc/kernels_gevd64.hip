// Order-64 fused subband update in float64 (BASELINE config 5: 64 loudspeakers x 128 control points), one workgroup of
// 16 waves per frequency bin.  Same stages as the other update kernels (reference Python/apvast.py:20-36, 329-364,
// 406-414; spec Matlab/ControlMethods/jdiag.m:103-117), arranged so that every dense product runs on the matrix cores
// and the eigen-iteration does its sweeps in float32:
//
//   stage 0  R_B, R_D = X^H X on v_mfma_f64_16x16x4_f64, one 16 x 16 tile per wave, operands straight from HBM/L2
//   stage 1  Cholesky of R_D + reg I in LDS; W = L^-1 in place                                   apvast.py:22-27
//   stage 2  C = W R_B W^H: two complex 64^3 products on the f64 MFMA                            apvast.py:28-29
//   stage 3  eigenvectors of C                                                                   apvast.py:30
//            (a) float32 BLOCK Jacobi: 8 blocks of 8, round-robin over the 28 block pairs (7 rounds of 4 pairs per
//                sweep).  A round solves its four 16 x 16 pair problems with the register-resident wave-level Jacobi of
//                the order-16 kernel (gevd16_common.h, one wave each; only the pairs between the two blocks, except in round 0
//                where the pairs inside every block are rotated too) and applies the four 16 x 16 unitary factors to
//                C and V as 16 x 16 x 16 complex products on v_mfma_f32_16x16x4_f32, one tile per wave: 7 barriers-pairs
//                per sweep instead of the 63 of a plain cyclic Jacobi.
//            (b) float64 refinement of that eigenvector matrix (Ogita & Aishima 2018), four complex 64^3 products on the
//                f64 MFMA per step, repeated until every correction |Z_ij| <= 3e-5 (what is then left is |Z|^2 <= 1e-9)
//   stage 4  sort                                                                                apvast.py:32-35
//   stage 5  X = W^H Q, one product                                                              apvast.py:31
//   stage 6  w_V = sum_{i<V} (x_i^H r)/(lam_i+mu) x_i                                            apvast.py:406-414
//
// LDS: two regions of 66 KB (R_B -> W R_B -> [C, V in float32] -> refinement work space; R_D -> L -> W -> V in float64).
// C and W wait in a slot of HBM scratch per bin (L2 resident) while the LDS is used for the sweeps: W as its ten 16 x 16 tiles on and
// below the diagonal, C in full.  (C as its lower tiles was built and measured in round 4: the refinement then reads a tile above
// the diagonal as the conjugate transpose of its mirror, and whichever way the summation index is dealt, the matrix cores want the
// row index of an operand on lane & 15 -- sixteen lanes striding over rows 1 KB apart: 24 KB less written per bin, 6 us more per
// refinement, 2.527 -> 2.588 ms per launch.  Both orientations of a tile are what the two waves that read it want: the full matrix.)
//
// Two kernels share the stages.  gevd64_kernel: one bin per workgroup.  gevd64x2_kernel: TWO bins per workgroup; the
// float64 stages run for one bin after the other, but the float32 sweeps of the two bins (66 KB each: both fit) are
// interleaved: while four waves solve the pair problems of one bin (latency bound, one wave per SIMD), the other twelve
// apply the previous round's factors of the other bin on the matrix cores (the six tiles of C above the diagonal of the pair
// grid, mirrored; the four diagonal tiles come rotated out of the pair solves themselves).
#include "apv_internal.h"

#include "gevd16_common.h"

namespace {

constexpr int N64 = 64;
constexpr int LDD = 65;            // row stride of a c128 matrix in LDS
constexpr int LDF = 66;            // row stride of a c64 matrix in LDS (rows stay 16-byte aligned)
constexpr int REGION = N64 * LDF * 8 * 2;          // bytes: two c64 matrices; >= one c128 matrix (64 * 65 * 16)
constexpr int NBLK = 8, BS = 8;                    // block Jacobi: 8 blocks of 8

using C128 = Cx<double>;
using C64 = Cx<float>;

// round-robin tournament of NBLK players, round r in [0, NBLK-1), slot a in [0, NBLK/2): the pair (P < Q)
__device__ __forceinline__ void rr_pair8(int r, int a, int& P, int& Q) {
    constexpr int m1 = NBLK - 1;
    int u, v;
    if (a == 0) {
        u = m1;
        v = r;
    } else {
        u = (r + a) % m1;
        v = (r - a + m1) % m1;
    }
    P = u < v ? u : v;
    Q = u < v ? v : u;
}

// one 16 x 16 tile (ti, tj) of a complex 64 x 64 x 64 product on the f64 MFMA; fa(i, k), fb(k, j) fetch operand elements;
// k runs over [k_begin, k_end) in steps of 4.  out element t is (16 ti + (lane >> 4) + 4 t, 16 tj + (lane & 15)).
// PREFETCH_A: the A operands of the NEXT group of sixteen k are fetched before this group's MFMAs (for A operands that come from
// global memory -- C and W in the scratch slot: one L2 round trip per product instead of one per group)
template <bool PREFETCH_A = false, typename FA, typename FB>
__device__ __forceinline__ void cmm64_tile(FA fa, FB fb, int ti, int tj, int lane, int k_begin, int k_end, C128 out[4]) {
    // three real products per k-step (Karatsuba): P1 = sum ar br, P2 = sum ai bi, P3 = sum (ar + ai)(br + bi);
    // re = P1 - P2, im = P3 - P1 - P2.  These phases are bound by the f64 matrix pipe: 12 MFMAs and two adds beat 16 MFMAs.
    d4 p1 = {0, 0, 0, 0}, p2 = {0, 0, 0, 0}, p3 = {0, 0, 0, 0};
    const int il = lane & 15, kq = lane >> 4;
    const int i = 16 * ti + il, j = 16 * tj + il;
    C128 an[4];
    if constexpr (PREFETCH_A) {
#pragma unroll
        for (int u = 0; u < 4; ++u) an[u] = fa(i, k_begin + 4 * u + kq);
    }
    for (int k0 = k_begin; k0 < k_end; k0 += 16) {
        C128 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (PREFETCH_A) a[u] = an[u];
            else a[u] = fa(i, k0 + 4 * u + kq);
            b[u] = fb(k0 + 4 * u + kq, j);
        }
        if constexpr (PREFETCH_A) {
            const int kn = (k0 + 16 < k_end) ? k0 + 16 : k0;          // (the last group fetches its own operands again: no branch)
#pragma unroll
            for (int u = 0; u < 4; ++u) an[u] = fa(i, kn + 4 * u + kq);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u].x, b[u].x, p1, 0, 0, 0);
            p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u].y, b[u].y, p2, 0, 0, 0);
            p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u].x + a[u].y, b[u].x + b[u].y, p3, 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) out[t] = mk<double>(p1[t] - p2[t], p3[t] - p1[t] - p2[t]);
}

__device__ __forceinline__ C128 cj(C128 w) { return mk<double>(w.x, -w.y); }

// sum of v over the workgroup (every thread gets it); red: 16 doubles of LDS
__device__ __forceinline__ double block_sum(double v, double* red, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w];
    return s;
}


// stage stamps (apv_debug_set_stamps; tools/probes/stage_stamps64.py): s_memtime of thread 0 at a phase boundary of bin k
__device__ __forceinline__ void stamp64(const GevdParams& p, bool z1, int k, int i) {
    if (p.stamps != nullptr && threadIdx.x == 0) p.stamps[((size_t)(z1 ? 1 : 0) * p.K + k) * 16 + i] = __builtin_amdgcn_s_memtime();
}

// ---- shared memory of a workgroup, handed to the stage functions ------------------------------------------------------
struct Sh {
    unsigned char* regA;
    unsigned char* regB;
    C128* sr;
    C128* scoef;
    double* sDinv;
    double* sLam;
    double (*sPart)[N64];
    double (*sPartI)[N64];
    double* sRed;
    int* sOrder;
    int* sFlag;
};

constexpr size_t SLOT_BYTES = (size_t)2 * N64 * N64 * 16 + (size_t)N64 * N64 * 8 + (size_t)N64 * 16;   // C, W (c128), one c64 matrix, r

// stages 0-2 for bin k: leaves C and W in the scratch slot, the scaled float32 copy of C at cf_dst (row stride cf_ld; LDS or
// scratch) and, if vf_dst is given, V = I beside it.  Returns the status (1: not positive definite), or -1 on a debug stop.
template <bool FUSED, typename XT>
__device__ __forceinline__ int front64(const GevdParams& p, const Sh& sh, bool z1, int k, C128* gC, C128* gW, C64* cf_dst, int cf_ld,
                                       C64* vf_dst, double& normF2, double& scl, bool cf_lower_only = false) {
    const XT* const pXB = reinterpret_cast<const XT*>(z1 ? p.XB1 : p.XB);
    const XT* const pXD = reinterpret_cast<const XT*>(z1 ? p.XD1 : p.XD);
    const XT* const pd = reinterpret_cast<const XT*>(z1 ? p.d1 : p.d);
    C128* const RA = reinterpret_cast<C128*>(sh.regA);
    C128* const RB = reinterpret_cast<C128*>(sh.regB);
    C128* const sr = sh.sr;
    double* const sRed = sh.sRed;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int il = lane & 15, kq = lane >> 4;
    const int ti = wave >> 2, tj = wave & 3;           // this wave's 16 x 16 tile of a 64 x 64 matrix
    int status = 0;
    normF2 = 0.0;
    scl = 1.0;

    // ---------------- stage 0 ----------------
    if constexpr (FUSED) {
        const int M = p.M;
        // Both slabs go through LDS, 16 control points of each per chunk: every thread of the workgroup fetches 16 contiguous bytes
        // of the chunk (whole lines, each slab element read from memory ONCE) while the matrix cores work on the chunk before, and
        // the sixteen waves take their operands from LDS.  (Round 2 had every wave fetch its own operands from L2: each element
        // eight times, in 128-byte pieces, and the loop waited for memory M / 16 times: 85 us per bin against 28 us of MFMA time.)
        // R is Hermitian: only the ten tiles on and above the diagonal are formed (round 3; sixteen before).  Waves 0-3 take the
        // diagonal tiles of BOTH matrices, three real products each per k-step (Re R = sum xr (x) xr + xi (x) xi, P = sum xr (x) xi,
        // Im R = P - P^T inside the tile); waves 4-15 one tile above the diagonal of ONE matrix, four products.  A SIMD hosts one
        // wave of the first kind and three of the second: 18 matrix instructions per k-step where the full matrices took 24 --
        // the stage is bound by the f64 matrix pipe (65 cycles an instruction, nothing co-issues with it).
        const bool dwave = wave < 4;
        const int oj = dwave ? 0 : wave - 4, om = oj >= 6 ? 1 : 0, ou = oj - 6 * om;        // off-diagonal job: matrix, tile 0..5
        const int ci = dwave ? wave : (ou >= 3) + (ou >= 5);                                  // (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
        const int cj_ = dwave ? wave : (ou < 3 ? ou + 1 : (ou < 5 ? ou - 1 : 3));
        const XT* XBk = pXB + (size_t)k * M * N64;
        const XT* XDk = pXD + (size_t)k * M * N64;
        const XT* dvk = pd + (size_t)k * M;
        constexpr int CH = 16;                                              // control points per chunk
        constexpr int CHUNK_ELEMS = CH * N64;                               // elements of one slab per chunk
        constexpr int VEC = 16 / (int)sizeof(XT);                           // elements per 16-byte fetch: 2 (c64) or 1 (c128)
        constexpr int FETCHES = 2 * CHUNK_ELEMS / VEC / 1024;               // 16-byte fetches per thread and chunk: 1 (c64) or 2 (c128)
        using V16 = __attribute__((ext_vector_type(4))) unsigned;
        // two chunk buffers [2 slabs][CH][64] XT each, in region A (R is written there only after the loop)
        XT* const sX = reinterpret_cast<XT*>(sh.regA);
        XT* const sD = sX + 2 * 2 * CHUNK_ELEMS;                            // [2][CH] target samples of the chunk
        d4 bre = {0, 0, 0, 0}, bp = {0, 0, 0, 0}, dre = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
        double rx = 0, ry = 0;
        const int n_chunks = (M + CH - 1) / CH;
        V16 stage[FETCHES];
        XT dstage;
        auto fetch = [&](int c) {
#pragma unroll
            for (int f = 0; f < FETCHES; ++f) {
                const int e = (f * 1024 + tid) * VEC;                       // element of the [2][CH][64] chunk pair
                const int slab = e / CHUNK_ELEMS, off = e - slab * CHUNK_ELEMS;
                const int m = c * CH + off / N64;
                const XT* src = (slab ? XDk : XBk) + (size_t)c * CHUNK_ELEMS + off;
                V16 v = {0u, 0u, 0u, 0u};
                if (m < M) v = *reinterpret_cast<const V16*>(src);          // rows are 64 elements: a 16-byte piece never straddles one
                stage[f] = v;
            }
            dstage.x = 0;
            dstage.y = 0;
            if (tid < CH && c * CH + tid < M) dstage = dvk[c * CH + tid];
        };
        auto stash = [&](int buf) {
#pragma unroll
            for (int f = 0; f < FETCHES; ++f)
                *reinterpret_cast<V16*>(sX + (size_t)buf * 2 * CHUNK_ELEMS + (size_t)(f * 1024 + tid) * VEC) = stage[f];
            if (tid < CH) sD[buf * CH + tid] = dstage;
        };
        fetch(0);
        for (int c = 0; c < n_chunks; ++c) {
            const int buf = c & 1;
            stash(buf);
            __syncthreads();
            if (c + 1 < n_chunks) fetch(c + 1);                             // in flight while the MFMAs below run
            const XT* cb = sX + (size_t)buf * 2 * CHUNK_ELEMS;
            const XT* cd = cb + CHUNK_ELEMS;
            if (dwave) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = (4 * u + kq) * N64;
                    const XT ba = cb[row + 16 * ci + il], da = cd[row + 16 * ci + il];
                    double ar = ba.x, ai = ba.y;
                    bre = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, ar, bre, 0, 0, 0);
                    bre = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, ai, bre, 0, 0, 0);
                    bp = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, ai, bp, 0, 0, 0);          // P = sum xr (x) xi
                    ar = da.x; ai = da.y;
                    dre = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, ar, dre, 0, 0, 0);
                    dre = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, ai, dre, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, ai, dp, 0, 0, 0);
                }
            } else {
                const XT* cs = om ? cd : cb;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = (4 * u + kq) * N64;
                    const XT xa = cs[row + 16 * ci + il], xb = cs[row + 16 * cj_ + il];
                    const double ar = xa.x, ai = xa.y, br = xb.x, bi = xb.y;
                    bre = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, bre, 0, 0, 0);
                    bre = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, bi, bre, 0, 0, 0);
                    bp = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, bp, 0, 0, 0);          // Im R = xr (x) xi - xi (x) xr
                    bp = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, br, bp, 0, 0, 0);
                }
            }
            // r = X_B^H d: wave w takes control point w of the chunk, lane l loudspeaker l; the sixteen partial sums meet below
            {
                const XT xv = cb[wave * N64 + lane], dm = sD[buf * CH + wave];
                rx = fma_t((double)xv.y, (double)dm.y, fma_t((double)xv.x, (double)dm.x, rx));       // conj(x) * d
                ry = fma_t(-(double)xv.y, (double)dm.x, fma_t((double)xv.x, (double)dm.y, ry));
            }
            // (the next pass writes the OTHER buffer; this one is written again two passes on, behind the next pass's barrier)
        }
        __syncthreads();                                                    // every wave is done with the chunk buffers (region A)
        // the sixteen partial sums of r meet in region A (the chunk buffers are spent, R is not there yet), in a fixed order
        {
            double (*const sR16)[N64] = reinterpret_cast<double(*)[N64]>(sh.regA);          // [32][64]: re, then im
            sR16[wave][lane] = rx;
            sR16[16 + wave][lane] = ry;
            __syncthreads();
            if (tid < N64) {
                double sx = 0, sy = 0;
#pragma unroll
                for (int w2 = 0; w2 < 16; ++w2) {
                    sx += sR16[w2][tid];
                    sy += sR16[16 + w2][tid];
                }
                sr[tid] = mk<double>(sx, sy);
            }
            __syncthreads();
        }
        // Diagonal tiles: Im R = P - P^T, P to LDS first (imaginary slot), the transposed element read back from the same tile.
        // The other waves hold a finished tile above the diagonal and write it twice: as it is, and conjugated to its mirror.
        if (dwave) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                RA[(16 * ci + kq + 4 * t) * LDD + 16 * ci + il] = mk<double>(bre[t], bp[t]);
                RB[(16 * ci + kq + 4 * t) * LDD + 16 * ci + il] = mk<double>(dre[t], dp[t]);
            }
        } else {
            C128* const R = om ? RB : RA;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                R[(16 * ci + kq + 4 * t) * LDD + 16 * cj_ + il] = mk<double>(bre[t], bp[t]);
                R[(16 * cj_ + il) * LDD + 16 * ci + kq + 4 * t] = mk<double>(bre[t], -bp[t]);
            }
        }
        __syncthreads();
        double pbt[4] = {0, 0, 0, 0}, pdt[4] = {0, 0, 0, 0};
        if (dwave) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                pbt[t] = RA[(16 * ci + il) * LDD + 16 * ci + kq + 4 * t].y;                   // P[col][row]
                pdt[t] = RB[(16 * ci + il) * LDD + 16 * ci + kq + 4 * t].y;
            }
        }
        __syncthreads();
        if (dwave) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                RA[(16 * ci + kq + 4 * t) * LDD + 16 * ci + il].y = bp[t] - pbt[t];
                RB[(16 * ci + kq + 4 * t) * LDD + 16 * ci + il].y = dp[t] - pdt[t];
            }
        }
    } else {
        const C128* gRB = reinterpret_cast<const C128*>(p.RB) + (size_t)k * N64 * N64;
        const C128* gRD = reinterpret_cast<const C128*>(p.RD) + (size_t)k * N64 * N64;
        #pragma unroll
        for (int idx4 = 0; idx4 < 4; ++idx4) {
            const int idx = tid + 1024 * idx4;
            const int i = idx >> 6, j = idx & 63;
            RA[i * LDD + j] = gRB[idx];
            RB[i * LDD + j] = gRD[idx];
        }
        if (tid < N64) sr[tid] = p.r ? reinterpret_cast<const C128*>(p.r)[(size_t)k * N64 + tid] : mk<double>(0, 0);
    }
    __syncthreads();

    stamp64(p, z1, k, 8);
    if (p.debug_stop == 1) return -1;

    // ---------------- stage 1: Cholesky of B + reg I (lower, in place), then W = L^-1 ----------------
    if (tid < N64) RB[tid * LDD + tid] = mk<double>(RB[tid * LDD + tid].x + p.reg_dark, 0);
    if (tid == 0) sh.sFlag[1] = 0;
    __syncthreads();
    // Blocked right-looking factorisation in 16 x 16 tiles.  The diagonal factor L_kk itself is never needed: the panel is
    // L_ik = R_ik W_kk^H with W_kk = L_kk^-1, the trailing update R_ij -= L_ik L_jk^H, the inverse below takes W_kk and the
    // off-diagonal tiles of L, and the whitening takes W only.  Per tile column:
    //   (a) ONE wave turns [D_kk | I] into [. | L1^-1] by elimination (rows in registers: lane = row i + 16 x column group holds
    //       D[i][4cq..] and the right half's [i][4cq..]; a step hands its pivot column and the pivot row of the right half round
    //       through 512 bytes of LDS -- no barrier, the LDS executes a wave's accesses in order) and scales row i by 1/sqrt(d_i):
    //       that IS W_kk (apvast.py:24: B = L L^H).  Its conjugate transpose goes to the UPPER triangle of the tile (the mirrored
    //       place, diagonal included): nothing reads the upper triangle of the region otherwise.
    //   (b) panel tiles and (c) trailing tiles on the f64 matrix cores, one tile per wave.
    // (Rounds 2-3a eliminated column by column over the whole matrix: 64 steps of 0.6 us, bound by the f64 vector ALU issuing
    // every wave's share of every step.)
    {
        C128* const gjbuf = reinterpret_cast<C128*>(&sh.sPartI[0][0]);          // [16] pivot column, [16] pivot row of the right half
        for (int kb = 0; kb < 4; ++kb) {
            const int o = 16 * kb;
            if (wave == 0) {
                const int i = lane & 15, cq = lane >> 4;
                double bx[4], by[4], wx[4], wy[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = 4 * cq + u;
                    const C128 v = (j <= i) ? RB[(o + i) * LDD + o + j] : cj(RB[(o + j) * LDD + o + i]);     // the lower triangle is current
                    bx[u] = v.x;
                    by[u] = (j == i) ? 0.0 : v.y;
                    wx[u] = (j == i) ? 1.0 : 0.0;
                    wy[u] = 0.0;
                }
                double dsc = 1.0;
                bool bad = false;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int qc = q >> 2, qu = q & 3;
                    if (cq == qc) gjbuf[i] = mk<double>(bx[qu], by[qu]);          // column q of the left half
                    if (i == q) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) gjbuf[16 + 4 * cq + u] = mk<double>(wx[u], wy[u]);      // row q of the right half
                    }
                    const double dq = gjbuf[q].x;
                    bad = bad || !(dq > 0.0) || !(dq < 1e300);
                    const double inv = rcp_full(dq);
                    if (i == q) dsc = rsq_full(dq);
                    if (i > q) {
                        const C128 ci = gjbuf[i];
                        const double mx = ci.x * inv, my = ci.y * inv;          // multiplier m = B[i][q] / d_q
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const C128 cjv = gjbuf[4 * cq + u], rw = gjbuf[16 + 4 * cq + u];
                            // left half: -= m conj(B[j][q]) (row q of the Hermitian left half); right half: -= m W[q][j]
                            bx[u] = fma_t(-my, cjv.y, fma_t(-mx, cjv.x, bx[u]));
                            by[u] = fma_t(mx, cjv.y, fma_t(-my, cjv.x, by[u]));
                            wx[u] = fma_t(my, rw.y, fma_t(-mx, rw.x, wx[u]));
                            wy[u] = fma_t(-my, rw.x, fma_t(-mx, rw.y, wy[u]));
                        }
                    }
                }
                if (bad && lane == 0) sh.sFlag[1] = 1;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (4 * cq + u <= i) RB[(o + 4 * cq + u) * LDD + o + i] = mk<double>(wx[u] * dsc, -(wy[u] * dsc));   // conj(W[o+i][o+j]) at [o+j][o+i]
            }
            __syncthreads();
            if (sh.sFlag[1] != 0) {                          // uniform
                status = 1;
                break;
            }
            if (kb == 3) break;
            // (b) panel: L_ik = R_ik W_kk^H for the tiles below; B operand W_kk^H[k'][j] = conj(W[j][k']) = RB[o + k'][o + j], k' <= j
            if (wave < 3 - kb) {
                const int ib = kb + 1 + wave;
                C128 lacc[4];
                cmm64_tile([&](int i, int kk) { return RB[i * LDD + kk]; },
                           [&](int kk, int j) { return kk <= j ? RB[kk * LDD + j] : mk<double>(0, 0); }, ib, kb, lane, o, o + 16, lacc);
#pragma unroll
                for (int t = 0; t < 4; ++t) RB[(16 * ib + kq + 4 * t) * LDD + o + il] = lacc[t];          // only this wave reads this tile
            }
            __syncthreads();
            // (c) trailing update of the tiles (i, j), kb < j <= i: R_ij -= L_ik L_jk^H
            {
                const int rem = 3 - kb, ntile = rem * (rem + 1) / 2;
                if (wave < ntile) {
                    int ib = kb + 1, w2 = wave;
                    while (w2 > ib - kb - 1) {                // tiles (kb+1,kb+1), (kb+2,kb+1), (kb+2,kb+2), ...
                        w2 -= ib - kb;
                        ++ib;
                    }
                    const int jb = kb + 1 + w2;
                    C128 uacc[4];
                    cmm64_tile([&](int i, int kk) { return RB[i * LDD + kk]; }, [&](int kk, int j) { return cj(RB[j * LDD + kk]); }, ib, jb,
                               lane, o, o + 16, uacc);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        C128* e = &RB[(16 * ib + kq + 4 * t) * LDD + 16 * jb + il];
                        const C128 v = *e;
                        *e = mk<double>(v.x - uacc[t].x, v.y - uacc[t].y);
                    }
                }
            }
            __syncthreads();
        }
    }
    stamp64(p, z1, k, 9);
    if (p.debug_stop == 2) return -1;
    if (status == 0) {
        // W = L^-1: the diagonal tiles are done (mirrored, conjugated).
        //   (2) the blocks below the diagonal, one block diagonal after the other: W_ba = -W_bb (sum_{a <= k < b} L_bk W_ka).  The
        //       accumulator of the inner sum is the B operand of the second product as it stands (register t is row 4 t + kq).
        auto Wel = [&](int i, int kk) { return cj(RB[kk * LDD + i]); };          // W[i][kk], kk <= i (the diagonal is real)
        for (int dgl = 1; dgl < 4; ++dgl) {
            if (wave < 4 - dgl) {
                const int a = wave, b = wave + dgl;
                C128 sacc[4];
                cmm64_tile([&](int i, int kk) { return RB[i * LDD + kk]; },
                           [&](int kk, int j) { return kk >= j ? Wel(kk, j) : mk<double>(0, 0); }, b, a, lane, 16 * a, 16 * b, sacc);
                d4 p1 = {0, 0, 0, 0}, p2 = {0, 0, 0, 0}, p3 = {0, 0, 0, 0};
                const int i = 16 * b + il;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int kk = 16 * b + 4 * t + kq;
                    const C128 w = kk <= i ? Wel(i, kk) : mk<double>(0, 0);
                    p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w.x, sacc[t].x, p1, 0, 0, 0);
                    p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(w.y, sacc[t].y, p2, 0, 0, 0);
                    p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(w.x + w.y, sacc[t].x + sacc[t].y, p3, 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {                                    // element (16 b + kq + 4 t, 16 a + il) of -W_bb S, conjugated
                    RB[(16 * a + il) * LDD + 16 * b + kq + 4 * t] = mk<double>(-(p1[t] - p2[t]), p3[t] - p1[t] - p2[t]);
                }
            }
            __syncthreads();
        }

        stamp64(p, z1, k, 10);
        if (p.debug_stop == 3) return -1;
        // ---------------- stage 2: C = W A W^H ----------------
        // W to the scratch slot first: the stores drain while the products below run.  Only the ten tiles on and below the
        // diagonal (zeros above the diagonal INSIDE the diagonal tiles): stage 5 reads nothing else
#pragma unroll
        for (int idx4 = 0; idx4 < 4; ++idx4) {
            const int idx = tid + 1024 * idx4;
            const int i = idx >> 6, j = idx & 63;
            if ((j >> 4) <= (i >> 4)) gW[idx] = j <= i ? Wel(i, j) : mk<double>(0, 0);
        }
        C128 acc[4];
        // W is lower triangular: W[i][k] = 0 for k > i (the upper triangle of the region still holds R_D)
        cmm64_tile([&](int i, int kk) { return kk <= i ? Wel(i, kk) : mk<double>(0, 0); },
                   [&](int kk, int j) { return RA[kk * LDD + j]; }, ti, tj, lane, 0, 16 * (ti + 1), acc);       // T = W A
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) RA[(16 * ti + kq + 4 * t) * LDD + 16 * tj + il] = acc[t];
        __syncthreads();
        // The length of this product's k-range goes with the COLUMN tile, and a SIMD hosts the waves of one value of wave & 3: with
        // the kernel's usual (ti, tj) = (wave >> 2, wave & 3) one SIMD would run four full-length tiles and another four of a quarter
        // the length.  Here the wave takes tile (wave & 3, wave >> 2): every SIMD gets one tile of each length (16 -> 10 k-blocks on
        // the busiest matrix pipe).
        const int ti2 = wave & 3, tj2 = wave >> 2;
        cmm64_tile([&](int i, int kk) { return RA[i * LDD + kk]; },
                   [&](int kk, int j) { return kk <= j ? cj(Wel(j, kk)) : mk<double>(0, 0); }, ti2, tj2, lane, 0, 16 * (tj2 + 1),
                   acc);                                                                                          // C = T W^H
        double nrm = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = 16 * ti2 + kq + 4 * t, col = 16 * tj2 + il;
            if (row == col) acc[t].y = 0;
            nrm += acc[t].x * acc[t].x + acc[t].y * acc[t].y;
            gC[row * N64 + col] = acc[t];
        }
        normF2 = block_sum(nrm, sRed, tid);          // (its barriers also end every read of T in region A)
        const int sexp = (normF2 > 0.0) ? -(ilogb(normF2) / 2) : 0;
        scl = ldexp(1.0, sexp);
        // float32 working copy: C scaled to ||C||_F ~ 1 (and V = I beside it); cf_lower_only: a copy that waits in the scratch slot
        // holds the tiles on and below the diagonal only (load_cf_lower mirrors them)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = 16 * ti2 + kq + 4 * t, col = 16 * tj2 + il;
            if (!cf_lower_only || ti2 >= tj2) cf_dst[row * cf_ld + col] = mk<float>((float)(acc[t].x * scl), (float)(acc[t].y * scl));
            if (vf_dst != nullptr) vf_dst[row * cf_ld + col] = mk<float>(row == col ? 1.f : 0.f, 0.f);
        }
    }
    __syncthreads();
    return status;
}

// ---- float32 block Jacobi: the tasks of a round ------------------------------------------------------------------------
// pair problem w of round r, solved by one wave: reads its 16 x 16 sub-matrix of Cf, leaves the unitary factor in U and writes
// the ROTATED sub-matrix back (the diagonal tile of the pair grid: the wave holds U_w^H C_ww U_w in its registers when the
// rotations are done, so the matrix cores never see those four tiles: 448 instead of 576 MFMAs per round).  Returns this
// lane's share of the off-diagonal weight of what it wrote.
__device__ __forceinline__ float inner_solve(C64* Cf, C64* U, int r, int w, int lane) {
    const int ua = lane >> 3, ub = lane & 7;                      // the 8 x 8 grid of 2 x 2 blocks of a pair problem
    int P, Q;
    rr_pair8(r, w, P, Q);
    auto idx = [&](int x) { return x < BS ? BS * P + x : BS * Q + (x - BS); };
    const int r0 = idx(ua), r1 = idx(8 + ua), c0 = idx(ub), c1 = idx(8 + ub);
    C64 tt = Cf[r0 * LDF + c0], tb = Cf[r0 * LDF + c1], bt = Cf[r1 * LDF + c0], bb = Cf[r1 * LDF + c1];
    C64 v0t = mk<float>((2 * ua == ub) ? 1.f : 0.f, 0.f), v0b = mk<float>((2 * ua == 8 + ub) ? 1.f : 0.f, 0.f);
    C64 v1t = mk<float>((2 * ua + 1 == ub) ? 1.f : 0.f, 0.f), v1b = mk<float>((2 * ua + 1 == 8 + ub) ? 1.f : 0.f, 0.f);
    // Round 0 of an outer sweep pairs every block once: a full inner sweep there covers the pairs INSIDE all eight
    // blocks; the other rounds rotate only the 64 pairs between their two blocks (the first 8 rounds of the
    // schedule, which leave the slots as they were).  Together: every one of the 2016 index pairs once per sweep.
    const bool full = (r == 0);
    if (full) jacobi16_sweep0<float, 15>(tt, tb, bt, bb, v0t, v0b, v1t, v1b, (float (*)[4]) nullptr, lane);
    else jacobi16_sweep0<float, 8>(tt, tb, bt, bb, v0t, v0b, v1t, v1b, (float (*)[4]) nullptr, lane);
    // where the slots ended up: after the full sweep top / bottom of slot u are the local indices 2 u, 2 u + 1; after the
    // eight cross rounds they are back at u, 8 + u
    const bool nat = full;
    const int it_b = nat ? 2 * ub : ub, ib_b = nat ? 2 * ub + 1 : 8 + ub;
    U[(2 * ua) * 17 + it_b] = v0t;
    U[(2 * ua) * 17 + ib_b] = v0b;
    U[(2 * ua + 1) * 17 + it_b] = v1t;
    U[(2 * ua + 1) * 17 + ib_b] = v1b;
    const int gt = idx(nat ? 2 * ua : ua), gb = idx(nat ? 2 * ua + 1 : 8 + ua), ht = idx(it_b), hb = idx(ib_b);
    if (gt == ht) tt.y = 0.f;                                     // the global diagonal stays real
    if (gb == hb) bb.y = 0.f;
    Cf[gt * LDF + ht] = tt;
    Cf[gt * LDF + hb] = tb;
    Cf[gb * LDF + ht] = bt;
    Cf[gb * LDF + hb] = bb;
    float off = tb.x * tb.x + tb.y * tb.y + bt.x * bt.x + bt.y * bt.y;
    if (gt != ht) off += tt.x * tt.x + tt.y * tt.y;
    if (gb != hb) off += bb.x * bb.x + bb.y * bb.y;
    return off;
}

// tile (a, b) of the pair grid of round r: C_ab <- U_a^H C_ab U_b on v_mfma_f32_16x16x4_f32; with `mirror` the Hermitian
// counterpart C_ba is written too.  Returns this lane's share of the off-diagonal weight of what it wrote.
__device__ __forceinline__ float outer_ctile(C64* Cf, const C64 (*sUr)[16 * 17], int r, int a, int b, int lane, bool mirror) {
    const int il = lane & 15, kq = lane >> 4;
    int Pa, Qa, Pb, Qb;
    rr_pair8(r, a, Pa, Qa);
    rr_pair8(r, b, Pb, Qb);
    auto ia = [&](int x) { return x < BS ? BS * Pa + x : BS * Qa + (x - BS); };
    auto ib = [&](int x) { return x < BS ? BS * Pb + x : BS * Qb + (x - BS); };
    const C64* Ua = sUr[a];
    const C64* Ub = sUr[b];
    // operands with k = 4 kq + s: four consecutive columns of the pair's index set (they stay inside one block)
    const int crow = ia(il), kc = ib(4 * kq);
    C64 tr[4], ubv[4], uav[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        tr[s] = Cf[crow * LDF + kc + s];
        ubv[s] = Ub[(4 * kq + s) * 17 + il];
        uav[s] = Ua[(4 * kq + s) * 17 + il];
    }
    f4 pre = {0, 0, 0, 0}, pim = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        pre = __builtin_amdgcn_mfma_f32_16x16x4f32(tr[s].x, ubv[s].x, pre, 0, 0, 0);
        pre = __builtin_amdgcn_mfma_f32_16x16x4f32(-tr[s].y, ubv[s].y, pre, 0, 0, 0);
        pim = __builtin_amdgcn_mfma_f32_16x16x4f32(tr[s].x, ubv[s].y, pim, 0, 0, 0);
        pim = __builtin_amdgcn_mfma_f32_16x16x4f32(tr[s].y, ubv[s].x, pim, 0, 0, 0);
    }
    // second product U_a^H (C_ab U_b): register s of the first accumulator is row 4 kq + s of the product, i.e. exactly the
    // B operand of k = 4 kq + s; A[i][k] = conj(U_a[k][i])
    f4 cre = {0, 0, 0, 0}, cim = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        cre = __builtin_amdgcn_mfma_f32_16x16x4f32(uav[s].x, pre[s], cre, 0, 0, 0);
        cre = __builtin_amdgcn_mfma_f32_16x16x4f32(uav[s].y, pim[s], cre, 0, 0, 0);
        cim = __builtin_amdgcn_mfma_f32_16x16x4f32(uav[s].x, pim[s], cim, 0, 0, 0);
        cim = __builtin_amdgcn_mfma_f32_16x16x4f32(-uav[s].y, pre[s], cim, 0, 0, 0);
    }
    // accumulator element t: row 4 kq + t, column il of the tile
    const int ccol = ib(il);
    float off = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = ia(4 * kq + t);
        C64 v = mk<float>(cre[t], cim[t]);
        if (row == ccol) v.y = 0.f;
        else off += v.x * v.x + v.y * v.y;
        Cf[row * LDF + ccol] = v;
        if (mirror) Cf[ccol * LDF + row] = mk<float>(v.x, -v.y);
    }
    return mirror ? 2.f * off : off;
}

// rows 16 rb .. 16 rb + 15 of V, columns of pair b of round r: V_b <- V_b U_b
__device__ __forceinline__ void outer_vtile(C64* Vf, const C64 (*sUr)[16 * 17], int r, int rb, int b, int lane) {
    const int il = lane & 15, kq = lane >> 4;
    int Pb, Qb;
    rr_pair8(r, b, Pb, Qb);
    auto ib = [&](int x) { return x < BS ? BS * Pb + x : BS * Qb + (x - BS); };
    const C64* Ub = sUr[b];
    const int vrow = 16 * rb + il, kc = ib(4 * kq);
    C64 vr[4], ubv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        vr[s] = Vf[vrow * LDF + kc + s];
        ubv[s] = Ub[(4 * kq + s) * 17 + il];
    }
    f4 vre = {0, 0, 0, 0}, vim = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        vre = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[s].x, ubv[s].x, vre, 0, 0, 0);
        vre = __builtin_amdgcn_mfma_f32_16x16x4f32(-vr[s].y, ubv[s].y, vre, 0, 0, 0);
        vim = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[s].x, ubv[s].y, vim, 0, 0, 0);
        vim = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[s].y, ubv[s].x, vim, 0, 0, 0);
    }
    const int ccol = ib(il);
#pragma unroll
    for (int t = 0; t < 4; ++t) Vf[(16 * rb + 4 * kq + t) * LDF + ccol] = mk<float>(vre[t], vim[t]);
}

// hand-over rule of the float sweeps: off-diagonal weight `off` (C scaled to ||C||_F^2 = nrm_s) small enough for the float64
// refinement, or no longer falling (float32 rounding)
__device__ __forceinline__ bool presolve_done(double off, double off_prev, double nrm_s, double tol) {
    return off <= tol * nrm_s || (off <= 1e-7 * nrm_s && off > 0.25 * off_prev);
}

// stage 3b .. 6 and the outputs of bin k; vf_src: the float32 eigenvector matrix of the sweeps (row stride vf_ld; LDS or scratch)
template <typename XT>
__device__ __forceinline__ void back64(const GevdParams& p, const Sh& sh, bool z1, int k, const C128* gC, const C128* gW,
                                       const C64* vf_src, int vf_ld, int status, int n_sweeps) {
    void* const pw = z1 ? p.w1 : p.w;
    void* const plam = z1 ? p.lam1 : p.lam;
    int32_t* const pstatus = z1 ? p.status1 : p.status;
    C128* const RA = reinterpret_cast<C128*>(sh.regA);
    C128* const RB = reinterpret_cast<C128*>(sh.regB);
    C128* const sr = sh.sr;
    C128* const scoef = sh.scoef;
    double* const sLam = sh.sLam;
    double (*const sPart)[N64] = sh.sPart;
    int* const sOrder = sh.sOrder;
    int* const sFlag = sh.sFlag;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int il = lane & 15, kq = lane >> 4;
    const int ti = wave >> 2, tj = wave & 3;
    int n_ref = 0;
    if (status == 0) {
        C128 acc[4];
        // ---------------- stage 3b: float64 refinement on the matrix cores ----------------
        #pragma unroll
        for (int idx4 = 0; idx4 < 4; ++idx4) {
            const int idx = tid + 1024 * idx4;
            const int i = idx >> 6, j = idx & 63;
            const C64 v = vf_src[i * vf_ld + j];
            RB[i * LDD + j] = mk<double>((double)v.x, (double)v.y);
        }
        __syncthreads();
        constexpr double kGuard2 = 9e-10;          // |Z_ij|^2 <= (3e-5)^2 on every pair: the step that meets it is the last
        constexpr double kClamp2 = 1e-2;           // |Z_ij| > 0.1 is outside the step's range: damped to 0.1
        bool converged = false;
        const int max_ref = 6;
        for (int it = 0; it < max_ref && !converged; ++it) {
            ++n_ref;
            C128 accT[4], accG[4], accS[4];
            // (the lane index goes through an empty asm once per step: the 64 global addresses of C's operands are loop invariant
            // otherwise, and the compiler computes them in front of the loop and parks them in scratch memory -- 28 of the
            // kernel's spilled registers)
            int lane_l = lane;
            asm volatile("" : "+v"(lane_l));
            // C is Hermitian: C[i][k] = conj(C[k][i]) read along a row of the scratch copy (coalesced)
            cmm64_tile<true>([&](int i, int kk) { return cj(gC[kk * N64 + i]); }, [&](int kk, int j) { return RB[kk * LDD + j]; }, ti, tj, lane_l, 0,
                       N64, accT);                                                                              // C V
            // (region A is free: the loop's last barrier ended every read of Z; T leaves the registers before the next product starts)
#pragma unroll
            for (int t = 0; t < 4; ++t) RA[(16 * ti + kq + 4 * t) * LDD + 16 * tj + il] = accT[t];
            cmm64_tile([&](int i, int kk) { return cj(RB[kk * LDD + i]); }, [&](int kk, int j) { return RB[kk * LDD + j]; }, ti, tj, lane, 0,
                       N64, accG);                                                                              // V^H V
            if (tid < 2) sFlag[tid] = 0;
            __syncthreads();
            cmm64_tile([&](int i, int kk) { return cj(RB[kk * LDD + i]); }, [&](int kk, int j) { return RA[kk * LDD + j]; }, ti, tj, lane, 0,
                       N64, accS);                                                                              // S = V^H C V
            if (ti == tj) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (kq + 4 * t == il) sLam[16 * ti + il] = accS[t].x / accG[t].x;                            // Rayleigh quotients
            }
            __syncthreads();
            C128 accZ[4];
            double lam2[4];
            bool bad = false;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = 16 * ti + kq + 4 * t, col = 16 * tj + il;
                const double di = sLam[row], dj = sLam[col];
                lam2[t] = 0.0;
                if (row == col) {
                    accZ[t] = mk<double>(0.5 * (1.0 - accG[t].x), 0.0);
                } else {
                    const double den = dj - di;
                    double zx = -0.5 * accG[t].x, zy = -0.5 * accG[t].y;        // equal quotients: no rotation inside the pair
                    if (den != 0.0) {
                        const double inv = 1.0 / den;
                        zx = __builtin_fma(-dj, accG[t].x, accS[t].x) * inv;
                        zy = __builtin_fma(-dj, accG[t].y, accS[t].y) * inv;
                    }
                    double z2 = zx * zx + zy * zy;
                    if (!(z2 <= kClamp2)) {
                        const double f = (z2 < 1e300) ? 0.1 * rsq_full(z2) : 0.0;
                        zx *= f;
                        zy *= f;
                        z2 = kClamp2;
                    }
                    bad = bad || !(z2 <= kGuard2);
                    accZ[t] = mk<double>(zx, zy);
                    const double gx = __builtin_fma(0.5, accG[t].x, zx), gy = __builtin_fma(0.5, accG[t].y, zy);
                    lam2[t] = -(gx * gx + gy * gy) * den;
                }
            }
            if (bad) sFlag[0] = 1;
            // second-order part of the eigenvalues: row sums over this tile's 16 columns, then over the four column tiles
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                double v = lam2[t];
                v += xcol<1>(v);
                v += xcol<2>(v);
                v += xcol<4>(v);
                v += xrow<1>(v, lane);
                if (il == 0) sPart[tj][16 * ti + kq + 4 * t] = v;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) RA[(16 * ti + kq + 4 * t) * LDD + 16 * tj + il] = accZ[t];              // every read of T is done
            __syncthreads();
            converged = (sFlag[0] == 0);
            C128 accV[4];
            cmm64_tile([&](int i, int kk) { return RB[i * LDD + kk]; }, [&](int kk, int j) { return RA[kk * LDD + j]; }, ti, tj, lane, 0, N64,
                       accV);                                                                                   // V Z
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const C128 v = RB[(16 * ti + kq + 4 * t) * LDD + 16 * tj + il];
                accV[t] = mk<double>(v.x + accV[t].x, v.y + accV[t].y);
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 4; ++t) RB[(16 * ti + kq + 4 * t) * LDD + 16 * tj + il] = accV[t];              // V''
            if (converged && tid < N64) sLam[tid] += sPart[0][tid] + sPart[1][tid] + sPart[2][tid] + sPart[3][tid];
            __syncthreads();
        }
        if (!converged) status = 2;
        stamp64(p, z1, k, 11);
        if (p.debug_stop == 6) {                       // profiling aid: sweeps and refinement steps this bin took
            if (pstatus != nullptr && tid == 0) pstatus[k] = 100 * n_sweeps + n_ref;
            return;
        }

        // ---------------- stage 4: descending order ----------------
        if (tid < N64) {
            const double li = sLam[tid];
            int rank = 0;
            for (int j = 0; j < N64; ++j) {
                const double lj = sLam[j];
                rank += (lj > li) || (lj == li && j < tid);
            }
            sOrder[rank] = tid;
        }
        // ---------------- stage 5: X = W^H Q   (W[k][i] = 0 for k < i) ----------------
        cmm64_tile<true>([&](int i, int kk) { return cj(gW[kk * N64 + i]); }, [&](int kk, int j) { return RB[kk * LDD + j]; }, ti, tj, lane, 16 * ti,
                   N64, acc);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) RA[(16 * ti + kq + 4 * t) * LDD + 16 * tj + il] = acc[t];
        __syncthreads();
        // ---------------- stage 6: coefficients (x_c^H r) / (lam_c + mu) ----------------
        if (tid < N64) {
            double sx = 0, sy = 0;
            for (int l = 0; l < N64; ++l) {
                const C128 v = RA[l * LDD + tid], rr = sr[l];
                sx += v.x * rr.x + v.y * rr.y;
                sy += v.x * rr.y - v.y * rr.x;
            }
            const double den = 1.0 / (sLam[tid] + p.mu);
            scoef[tid] = mk<double>(sx * den, sy * den);
        }
        __syncthreads();
    }


    // ---------------- outputs ----------------
    // w_V = sum over the V leading columns (descending order) of coef_c x_c: running sums over the sorted columns, in segments
    // of four held by sixteen threads per row (region B is free by now: V has gone into X); a rank's filter is then the totals of
    // the segments in front of its last column plus the running sum inside that segment.  (Rounds 2-3a walked the sorted
    // columns one after the other in 64 threads, every step an index read and two dependent reads behind it: 8 us.)
    C128* const Tp = RB;
    if (status != 1) {
        const int l = tid & 63, part = tid >> 6;
        double ax = 0, ay = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = sOrder[4 * part + u];
            const C128 cf = scoef[c], v = RA[l * LDD + c];
            ax += cf.x * v.x - cf.y * v.y;
            ay += cf.x * v.y + cf.y * v.x;
            Tp[l * LDD + 4 * part + u] = mk<double>(ax, ay);
        }
    }
    __syncthreads();
    for (int idx = tid; idx < p.nV * N64; idx += 1024) {
        const int t = idx >> 6, l = idx & 63;
        const int V = p.ranks[t];
        double ax = 0, ay = 0;
        if (status != 1 && V > 0) {
            const int seg = (V - 1) >> 2;
            for (int q = 0; q < seg; ++q) {
                const C128 v = Tp[l * LDD + 4 * q + 3];
                ax += v.x;
                ay += v.y;
            }
            const C128 v = Tp[l * LDD + V - 1];
            ax += v.x;
            ay += v.y;
        }
        const size_t o = ((size_t)k * p.nV + t) * N64 + l;
        if (p.out_c128) reinterpret_cast<double2*>(pw)[o] = make_double2(ax, ay);
        else reinterpret_cast<float2*>(pw)[o] = make_float2((float)ax, (float)ay);
    }
    if (tid < N64 && plam != nullptr) {
        const double lv = (status != 1) ? sLam[sOrder[tid]] : 0.0;
        if (p.out_c128) reinterpret_cast<double*>(plam)[(size_t)k * N64 + tid] = lv;
        else reinterpret_cast<float*>(plam)[(size_t)k * N64 + tid] = (float)lv;
    }
    if (p.U != nullptr) {
        C128* U = reinterpret_cast<C128*>(p.U) + (size_t)k * N64 * N64;
        #pragma unroll
        for (int idx4 = 0; idx4 < 4; ++idx4) {
            const int idx = tid + 1024 * idx4;
            const int i = idx >> 6, j = idx & 63;
            U[idx] = (status != 1) ? RA[i * LDD + sOrder[j]] : mk<double>(0, 0);
        }
    }
    if (pstatus != nullptr && tid == 0) pstatus[k] = status;
}


// ---- one bin per workgroup ---------------------------------------------------------------------------------------------
// XT: element type of the fused input slabs (float2 = c64, double2 = c128); FUSED = false takes explicit R_B, R_D, r (c128)
template <bool FUSED, typename XT>
__global__ void __launch_bounds__(1024) gevd64_kernel(const GevdParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char sRegA[REGION];
    __shared__ __attribute__((aligned(16))) unsigned char sRegB[REGION];
    __shared__ C128 sr[N64], scoef[N64];
    __shared__ double sDinv[N64], sLam[N64], sPart[4][N64], sPartI[4][N64], sRed[16];
    __shared__ int sOrder[N64];
    __shared__ int sFlag[2];
    __shared__ C64 sU[4][16 * 17];                     // the four 16 x 16 unitary factors of a block round
    const Sh sh{sRegA, sRegB, sr, scoef, sDinv, sLam, sPart, sPartI, sRed, sOrder, sFlag};
    const bool z1 = (blockIdx.y == 1);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ti = wave >> 2, tj = wave & 3;
    const int k = blockIdx.x;
    unsigned char* const slot = reinterpret_cast<unsigned char*>(p.Lspill) + ((size_t)blockIdx.y * p.K + k) * SLOT_BYTES;
    C128* const gC = reinterpret_cast<C128*>(slot);
    C128* const gW = gC + N64 * N64;
    C64* const Cf = reinterpret_cast<C64*>(sRegA);
    C64* const Vf = Cf + N64 * LDF;
    double normF2, scl;
    const int status = front64<FUSED, XT>(p, sh, z1, k, gC, gW, Cf, LDF, Vf, normF2, scl);
    if (status < 0 || p.debug_stop == 4) return;
    int n_sweeps = 0;
    if (status == 0) {
        // ---------------- stage 3a: float32 block Jacobi, all sixteen waves on the one bin ----------------
        const int max_sweeps = p.max_sweeps > 0 ? p.max_sweeps : 14;
        // off-diagonal weight at which the float sweeps hand over to the float64 refinement (a refinement step costs 0.6 of a sweep)
        const double kPreTol = p.sweep_tol2 < 0.0 ? -p.sweep_tol2 : 1e-8;
        double off_prev = 1e300;
        bool pre_done = false;
        for (int sweep = 0; sweep < max_sweeps && !pre_done; ++sweep) {
            ++n_sweeps;
            float offw = 0.f;
            for (int r = 0; r < NBLK - 1; ++r) {
                float od = 0.f;
                if (wave < 4 && p.debug_stop != 7) od = inner_solve(Cf, sU[wave], r, wave, lane);   // debug_stop 7 / 8: timing aids
                __syncthreads();
                if (p.debug_stop != 8) {
                    float o = 0.f;
                    if (ti != tj) o = outer_ctile(Cf, sU, r, ti, tj, lane, false);      // the diagonal tiles are the pair solves' own
                    outer_vtile(Vf, sU, r, ti, tj, lane);
                    if (r == NBLK - 2) offw = o + od;      // the last round of the sweep rewrites all of C: its off-diagonal weight
                }
                __syncthreads();
            }
            const double off = block_sum((double)offw, sRed, tid);
            pre_done = presolve_done(off, off_prev, normF2 * scl * scl, kPreTol);
            off_prev = off;
        }
    }
    if (p.debug_stop == 5 || p.debug_stop == 7 || p.debug_stop == 8) {
        int32_t* const pstatus = z1 ? p.status1 : p.status;
        if (pstatus != nullptr && tid == 0) pstatus[k] = 100 * n_sweeps;
        return;
    }
    back64<XT>(p, sh, z1, k, gC, gW, Vf, LDF, status, n_sweeps);
}

// ---- two bins per workgroup: float32 sweeps of the two bins interleaved ----------------------------------------------
__constant__ signed char kUpperA[6] = {0, 0, 0, 1, 1, 2};          // the six tiles above the diagonal of the 4 x 4 pair grid
__constant__ signed char kUpperB[6] = {1, 2, 3, 2, 3, 3};

template <bool FUSED, typename XT>
__global__ void __launch_bounds__(1024) gevd64x2_kernel(const GevdParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char sRegA[REGION];
    __shared__ __attribute__((aligned(16))) unsigned char sRegB[REGION];
    __shared__ C128 sr[N64], scoef[N64];
    __shared__ double sDinv[N64], sLam[N64], sPart[4][N64], sPartI[4][N64], sRed[16];
    __shared__ int sOrder[N64];
    __shared__ int sFlag[2];
    __shared__ C64 sU[2][4][16 * 17];                  // per bin: the four unitary factors of its round in flight
    __shared__ float sOffW[2][16];
    __shared__ float sOffD[2][4];                      // per bin: off-diagonal weight inside the four diagonal tiles (from the pair solves)
    const Sh sh{sRegA, sRegB, sr, scoef, sDinv, sLam, sPart, sPartI, sRed, sOrder, sFlag};
    const bool z1 = (blockIdx.y == 1);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int k0 = 2 * blockIdx.x;
    const int nb = (k0 + 1 < p.K) ? 2 : 1;
    // per-bin state lives in scalars selected by a uniform index (indexed local arrays would go to scratch memory)
    unsigned char* const slot0 = reinterpret_cast<unsigned char*>(p.Lspill) + ((size_t)blockIdx.y * p.K + k0) * SLOT_BYTES;
    unsigned char* const slot1 = slot0 + (nb == 2 ? SLOT_BYTES : 0);
    auto gCb = [&](int b) { return reinterpret_cast<C128*>(b ? slot1 : slot0); };
    auto gWb = [&](int b) { return gCb(b) + N64 * N64; };
    auto gFb = [&](int b) { return reinterpret_cast<C64*>(gCb(b) + 2 * N64 * N64); };
    auto gRb = [&](int b) { return reinterpret_cast<C128*>(gFb(b) + N64 * N64); };          // r = X_B^H d of the bin
    auto Cfb = [&](int b) { return reinterpret_cast<C64*>(b ? sRegB : sRegA); };
    auto Vfb = [&](int b) { return Cfb(b) + N64 * LDF; };
    int status0 = 0, status1 = 1;
    double normF2_0 = 0, normF2_1 = 0, scl0 = 1, scl1 = 1;
    stamp64(p, z1, k0, 0);
    // float64 front stages, one bin after the other.  The scaled float32 copy of the FIRST bin's C waits in the scratch slot while the
    // second bin's front stages use all of the LDS (its tiles on and below the diagonal only); the second bin's copy goes straight
    // to where its sweeps want it (region B is free once front64's last product has been summed)
    status0 = front64<FUSED, XT>(p, sh, z1, k0, gCb(0), gWb(0), nb == 2 ? gFb(0) : Cfb(0), nb == 2 ? N64 : LDF, nb == 2 ? nullptr : Vfb(0),
                                 normF2_0, scl0, nb == 2);
    if (status0 < 0) return;
    if (tid < N64) gRb(0)[tid] = sr[tid];
    __syncthreads();
    stamp64(p, z1, k0, 1);
    if (nb == 2) {
        status1 = front64<FUSED, XT>(p, sh, z1, k0 + 1, gCb(1), gWb(1), Cfb(1), LDF, Vfb(1), normF2_1, scl1);
        if (tid < N64) gRb(1)[tid] = sr[tid];
        __syncthreads();
        if (status0 == 0) {
            C64* const Cf = Cfb(0);
            C64* const Vf = Vfb(0);
            const C64* const src = gFb(0);
            #pragma unroll
            for (int idx4 = 0; idx4 < 4; ++idx4) {
                const int idx = tid + 1024 * idx4;
                const int i = idx >> 6, j = idx & 63;
                Vf[i * LDF + j] = mk<float>(i == j ? 1.f : 0.f, 0.f);
                if ((j >> 4) <= (i >> 4)) {                      // a stored tile; the tiles below the diagonal also fill their mirrors
                    const C64 v = src[idx];
                    Cf[i * LDF + j] = v;
                    if ((j >> 4) < (i >> 4)) Cf[j * LDF + i] = mk<float>(v.x, -v.y);
                }
            }
        }
    }
    __syncthreads();
    stamp64(p, z1, k0, 2);
    // ---------------- stage 3a for both bins: in every step four waves solve the pair problems of one bin's next round
    // (and write the rotated diagonal tiles back) while the other twelve apply the factors of the other bin's current round (the 6
    // tiles of C above the diagonal with their mirrors + 16 tiles of V = 28 products of 16^3); the bins swap roles from step to step
    enum { INNER = 0, OUTER = 1, DONE = 2 };
    int st0 = status0 == 0 ? INNER : DONE, st1 = (nb == 2 && status1 == 0) ? INNER : DONE;
    int rnd0 = 0, rnd1 = 0, swp0 = 0, swp1 = 0;
    double offp0 = 1e300, offp1 = 1e300;
    const int max_sweeps = p.max_sweeps > 0 ? p.max_sweeps : 14;
    const double kPreTol = p.sweep_tol2 < 0.0 ? -p.sweep_tol2 : 1e-8;
    // probe (apv_debug_set_stamps): cycles waves 0 (pair solves), 4 (a C tile + a V tile) and 15 (three V tiles) work per step, and
    // the steps' whole length with the barrier: slots 12 / 13 / 14 of the first bin's row, the latter in slot 15
    const bool dbg_t = (p.stamps != nullptr) && lane == 0 && (wave == 0 || wave == 4 || wave == 15);
    unsigned long long t_work = 0, t_all = 0;
    // the pair solves are one long dependent chain per step and every other wave of the step waits for them at the barrier:
    // their waves go first whenever they can issue
    if (wave < 4) __builtin_amdgcn_s_setprio(3);
    for (int step = 0; step < 2 * (NBLK - 1) * max_sweeps + 4 && !(st0 == DONE && st1 == DONE); ++step) {
        const int bS = step & 1, bU = bS ^ 1;
        const int stS = bS ? st1 : st0, stU = bU ? st1 : st0;
        const int rS = bS ? rnd1 : rnd0, rU = bU ? rnd1 : rnd0;
        const bool do_inner = (stS == INNER), do_outer = (stU == OUTER);
        const bool sweep_end = do_outer && rU == NBLK - 2;
        float offw = 0.f;
        const unsigned long long tw0 = dbg_t ? __builtin_amdgcn_s_memtime() : 0ull;
        if (wave < 4) {
            if (do_inner) {
                const float od = inner_solve(Cfb(bS), sU[bS][wave], rS, wave, lane);
                if (rS == NBLK - 2) {                            // the sweep's last round: what it leaves inside the diagonal tiles
                    const float w = wave_sum(od);
                    if (lane == 0) sOffD[bS][wave] = w;
                }
            }
        } else if (do_outer) {
            // 6 tiles of C above the diagonal (two products each, mirrored) + 16 tiles of V (one product) = 28 products on twelve
            // waves, seven per SIMD: waves 4-7 a C tile and a V tile, 8-9 a C tile, 10-15 two V tiles
            const int u = wave - 4;
            C64* const Cf = Cfb(bU);
            C64* const Vf = Vfb(bU);
            if (u < 6) {
                offw = outer_ctile(Cf, sU[bU], rU, kUpperA[u], kUpperB[u], lane, true);
                if (u < 4) outer_vtile(Vf, sU[bU], rU, 0, u, lane);
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int vt = 4 + 2 * (u - 6) + q;
                    outer_vtile(Vf, sU[bU], rU, vt >> 2, vt & 3, lane);
                }
            }
        }
        if (sweep_end) {
            const float w = wave_sum(offw);
            if (lane == 0) sOffW[step & 1][wave] = w;
        }
        if (dbg_t) {
            __builtin_amdgcn_s_waitcnt(0);                       // the wave's LDS stores have landed: its work is done
            t_work += __builtin_amdgcn_s_memtime() - tw0;
        }
        __syncthreads();
        if (dbg_t) t_all += __builtin_amdgcn_s_memtime() - tw0;
        if (do_inner) {
            if (bS) st1 = OUTER; else st0 = OUTER;
        }
        if (do_outer) {
            int nst = INNER, nr = rU + 1;
            if (sweep_end) {
                double off = 0;
#pragma unroll
                for (int w = 4; w < 10; ++w) off += (double)sOffW[step & 1][w];
#pragma unroll
                for (int w = 0; w < 4; ++w) off += (double)sOffD[bU][w];
                const int sw = (bU ? swp1 : swp0) + 1;
                const double nrm_s = bU ? normF2_1 * scl1 * scl1 : normF2_0 * scl0 * scl0;
                nst = (presolve_done(off, bU ? offp1 : offp0, nrm_s, kPreTol) || sw >= max_sweeps) ? DONE : INNER;
                nr = 0;
                if (bU) { swp1 = sw; offp1 = off; } else { swp0 = sw; offp0 = off; }
            }
            if (bU) { st1 = nst; rnd1 = nr; } else { st0 = nst; rnd0 = nr; }
        }
    }
    if (wave < 4) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    if (dbg_t) {
        unsigned long long* row = p.stamps + ((size_t)(z1 ? 1 : 0) * p.K + k0) * 16;
        row[wave == 0 ? 12 : (wave == 4 ? 13 : 14)] = t_work;
        if (wave == 0) row[15] = t_all;
    }
    stamp64(p, z1, k0, 3);
    // the SECOND bin's float32 eigenvector matrix waits in its scratch slot while the first bin's back stages use all of the LDS;
    // the first bin's is consumed from region A where it lies (back64 reads it before it writes anything there)
    if (nb == 2 && status1 == 0) {
        const C64* const Vf = Vfb(1);
        C64* const dst = gFb(1);
        #pragma unroll
        for (int idx4 = 0; idx4 < 4; ++idx4) {
            const int idx = tid + 1024 * idx4;
            dst[idx] = Vf[(idx >> 6) * LDF + (idx & 63)];
        }
    }
    __syncthreads();
    if (p.debug_stop == 5) {
        int32_t* const pstatus = z1 ? p.status1 : p.status;
        if (pstatus != nullptr && tid < nb) pstatus[k0 + tid] = 100 * (tid ? swp1 : swp0);
        return;
    }
    if (tid < N64) sr[tid] = gRb(0)[tid];
    __syncthreads();
    stamp64(p, z1, k0, 4);
    back64<XT>(p, sh, z1, k0, gCb(0), gWb(0), Vfb(0), LDF, status0, swp0);
    stamp64(p, z1, k0, 5);
    if (nb == 2) {
        __syncthreads();
        if (tid < N64) sr[tid] = gRb(1)[tid];
        __syncthreads();
        back64<XT>(p, sh, z1, k0 + 1, gCb(1), gWb(1), gFb(1), N64, status1, swp1);
    }
    stamp64(p, z1, k0, 6);
}

}  // namespace

size_t apv_gevd64_slot_bytes() { return SLOT_BYTES; }

// the conditions of apv_launch_gevd64 that are known when a handle is created (the arithmetic and fused / explicit are not:
// a float32 handle reaches this kernel through its fused entry point)
bool apv_gevd64_eligible(int n, int reg_mode, double reg_bright, double sweep_tol2) {
    static const bool off = (getenv("APV_NO_GEVD64") != nullptr);
    return !off && n == 64 && reg_mode == APV_REG_ABS && reg_bright == 0.0 && !(sweep_tol2 > 0.0);
}

// hipErrorNotSupported when the problem does not qualify (order != 64, float32 arithmetic, relative or bright loading, a
// caller-set sweep tolerance): the LDS kernel of kernels_gevd.hip then takes it
hipError_t apv_launch_gevd64(const GevdParams& p, int compute_dtype, bool fused, hipStream_t s) {
    static const bool off = (getenv("APV_NO_GEVD64") != nullptr);          // A/B switch: the LDS kernel
    static const bool single = (getenv("APV_GEVD64_SINGLE") != nullptr);   // A/B switch: one bin per workgroup
    // float32 arithmetic asked for at order 64 gets this kernel too when the inputs are the fused slabs: it is 1.4x as fast
    // as the float LDS kernel and more accurate than asked (explicit float32 statistics still go to the LDS kernel)
    if (off || !apv_gevd64_eligible(p.n, p.reg_mode, p.reg_bright, p.sweep_tol2) || (compute_dtype != APV_F64 && !fused) ||
        /* sweep_tol2 < 0: tuning aid, -value = hand-over threshold of the pre-solve */
        p.Lspill == nullptr)
        return hipErrorNotSupported;
    if (p.K <= 0) return hipSuccess;
    const bool xd = fused && p.x_c128;
    // one bin per workgroup while that still gives every CU its own workgroup (the two-bin kernel halves the grid), and for the
    // timing aids of the single-bin kernel (debug_stop 6, 7, 8)
    if (single || p.K * (p.n_zones > 1 ? 2 : 1) < 512 || p.debug_stop >= 6) {
        const dim3 grid(p.K, p.n_zones > 1 ? 2 : 1);
        if (xd) hipLaunchKernelGGL((gevd64_kernel<true, double2>), grid, dim3(1024), 0, s, p);
        else if (fused) hipLaunchKernelGGL((gevd64_kernel<true, float2>), grid, dim3(1024), 0, s, p);
        else hipLaunchKernelGGL((gevd64_kernel<false, float2>), grid, dim3(1024), 0, s, p);
    } else {
        const dim3 grid((p.K + 1) / 2, p.n_zones > 1 ? 2 : 1);
        if (xd) hipLaunchKernelGGL((gevd64x2_kernel<true, double2>), grid, dim3(1024), 0, s, p);
        else if (fused) hipLaunchKernelGGL((gevd64x2_kernel<true, float2>), grid, dim3(1024), 0, s, p);
        else hipLaunchKernelGGL((gevd64x2_kernel<false, float2>), grid, dim3(1024), 0, s, p);
    }
    return hipGetLastError();
}
