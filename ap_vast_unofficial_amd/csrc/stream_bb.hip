// Broadband (time-domain) AP-VAST on the device, float64: the reference's own algorithm, stage by stage.
//
//   apv_bb_init          allocate state, upload RIRs                       reference Python/apvast.py:97-151
//   apv_bb_process_block one hop                                            apvast.py:153-165
//     1 RIR convolution into the response rings                             apvast.py:167-194
//     2 WOLA of target / response rings (weights = 1), head of the overlap
//       buffers appended to the statistics rings                            apvast.py:197-311
//     3 R = sum_m Y_m Y_m^T, r = sum_m Y_m d_m from the statistics rings,
//       Y the data matrix of apvast.py:334-338 INCLUDING the sample that
//       scipy.linalg.toeplitz drops                 apvast.py:329-364
//     4 jdiag + rank-accumulated filters (apv_gevd_large)                   apvast.py:378-414
//     5 filter spectra = rfft(taps, N)                                      apvast.py:417-422
//     6 input spectra x filter spectra, inverse STFT, window, overlap-add   apvast.py:428-506
//
// Channel order on the device: c = m*L + l (as in stream.hip).  Everything is double: the dark matrix is loaded
// with an ABSOLUTE 1e-7 (apvast.py:23) against entries of order 1e-3, so float statistics would not survive.
#include "apv_internal.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

struct apv_bb {
    int N, H, K, L, M, C, P, J, S, V, n, zones, pad;      // V = number of solutions (ranks kept)
    int dialect, skip, ncols, toff, rel_loading, rel_dark_py;          // statistics conventions of the dialect
    long not_converged;    // hops whose joint diagonalisation hit the sweep cap
    int* d_ranks;          // [V] ascending
    int max_rank;          // the largest of them: the eigenpairs a hop consumes (apvast.py:406-414)
    int full_valid;        // U / lam hold ALL eigenpairs of the last hop (0: the leading block only; the rest is computed when read)
    double* nrm;           // [4] ||R_q||_2 for the relative loading
    int ring_off, stat_off, cur;
    int n_out;
    double* rir[2];        // [P][C]
    double* trir[2];       // [P][M]
    double* xhist[2][2];   // [buf][signal][P-1+H+pad]
    double* xin;           // [2][H]
    double* resp[4];       // [C][N] rings
    double* tresp[2];      // [M][N] rings
    double* inblk;         // [2][N] rings
    double* spec;          // scratch spectra [max(C, n_out)][K] c128
    double* ov[4];         // [C][N] overlap buffers of the weighted responses (linear, shifted by the kernel)
    double* tov[2];        // [M][N]
    double* stats[4];      // [C][S] rings
    double* tstats[2];     // [M][S] rings
    double* R;             // [4][n][n]: bright AA, BB, then dark AB, BA (zone A = (0, 2), zone B = (1, 3))
    double* r;             // [2][n]
    double* U;             // [2][n][n]
    double* lam;           // [2][n]
    double* w;             // [2][V][n]
    double* fspec;         // [n_out][K] c128 filter spectra (zone A ranks, zone B ranks, target A, target B)
    double* inspec;        // [2][K] c128
    double* outov;         // [n_out][N]
    double* out;           // [n_out][H]
    // perceptual weighting (off when nch == 0)
    int nch, norm_mode;
    double Cs, Ca, Leff;
    double* G2;            // [K][nch]
    double* G2T;           // [nch][K]
    double* tspec[2];      // [M][K] c128 target spectra (kept: the curves come from both zones before any scaling)
    // Round 4: the four response sets and the two target sets of a hop are ONE set of 4 C + 2 M channels -- rings, overlap buffers,
    // statistics rings and spectra are each one allocation, paths first, then the targets (resp[p], tresp[z], ov[p], ... point into
    // them) -- so that the hop's windowed transforms, its synthesis and its append to the statistics rings are one launch each
    // instead of six (apvast.py:197-311 runs the same three steps per buffer).
    double *resp_all, *ov_all, *stats_all, *pspec_all;
    // pinned staging of the per-hop call (round 4): the caller's arrays are pageable, and an asynchronous copy from / to pageable
    // memory is staged by the runtime behind a synchronisation of its own -- 83 us between the last kernel of a hop and the start
    // of its copy back in the trace.  The hop's inputs are copied here first and its outputs land here.
    double* pin_in;        // [2][H]
    double* pin_out;       // [n_out][H]
    int n_all;             // 4 C + 2 M
    double* Wgt[2];        // [M][K]
    // apv_bb_process_signal: G consecutive hops share ONE batched joint diagonalisation (allocated on first use)
    int grp;               // hops per group the buffers below are sized for (0: not allocated)
    double* g_xin;         // [grp][2][H] the group's input hops
    double* g_RA;          // [grp * nz][n][n] bright matrices, (hop, zone that runs) major
    double* g_RB;          // [grp * nz][n][n] dark matrices
    double* g_U;           // [grp * nz][n][n]
    double* g_lam;         // [grp * nz][n]
    double* g_r;           // [grp * nz][n]
    double* g_w;           // [grp * nz][V][n]
    double* g_nrm;         // [grp][4] + [grp * nz] dark norms in batch order
    double* g_inspec;      // [grp][2][K] c128
    double* g_out;         // [2 sets] a group's output hops as the caller's array holds them: [n_out / L][grp H][L] with
                           // cfg.out_layout = 1, else [grp][n_out][H]: ONE copy per group to the host
    double* g_pin;         // [2 sets] the same, page-locked: staging for a caller's pageable array (allocated when first needed)
    size_t g_pin_cap;      // doubles per set g_pin holds
    double* g_pin_in;      // [2 sets][grp][2][H] page-locked: a group's input hops, gathered here and sent in ONE copy
    int out_group;         // cfg.out_layout = 1: L (sample-major emit, see bb_back); 0: channel-major
    // (each of the above twice: the front stages of group g + 1 fill one set while the batched solve of group g reads the other)
    double* spec_out;      // [n_out][K] c128: the output stage's own spectra (the front stages use `spec` at the same time)
    hipStream_t front;     // the front stages of apv_bb_process_signal
    hipStream_t copy;      // a group's outputs on their way to the host while the next group is solved
    hipStream_t front2;    // the statistics of hop h beside the K1 / WOLA chain of hop h + 1 (apv_bb_process_signal)
    hipEvent_t ev_ring, ev_stat;             // hop's statistics rings written / read
    hipEvent_t ev_front[2], ev_back[2];      // group set filled / group set read for the last time
    hipEvent_t ev_out[2];                    // a group's outputs have reached g_pin
};

namespace {

#define BCHK(h, call)                                                                    \
    do {                                                                                 \
        hipError_t _e = (call);                                                          \
        if (_e != hipSuccess) return apv_fail(h, APV_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)

// true when [p, p + bytes) lies in page-locked host memory: a device-to-host copy into it is a DMA that runs asynchronously; any
// other host pointer is staged by the runtime and blocks the caller.  Blocks of apv_host_alloc are known without asking the
// runtime; with `ask` any other pointer is looked up there (a search and, for pageable memory, an error path: once per whole
// signal, never per hop).
bool host_is_pinned(const void* p, size_t bytes, bool ask) {
    if (apv_host_block_contains(p, bytes)) return true;
    if (!ask) return false;
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// dst[r][0 : width] = src[r][0 : width], r < rows (pitches in doubles), shared out over a few threads: the collection of a group
// of hops from the staging buffer into the caller's pageable array runs at memory speed, not at one core's
void copy_rows(double* dst, size_t dpitch, const double* src, size_t spitch, size_t width, size_t rows) {
    const size_t total = width * rows * sizeof(double);
    const unsigned hw = std::thread::hardware_concurrency();
    size_t nt = total >> 21;                            // a thread per 2 MiB
    nt = nt < 1 ? 1 : (nt > 8 ? 8 : nt);
    if (hw && nt > hw) nt = hw;
    // rows are split in pieces when there are fewer rows than threads
    const size_t pieces = rows >= nt ? 1 : (nt + rows - 1) / rows;
    const size_t units = rows * pieces, pw = (width + pieces - 1) / pieces;
    auto work = [&](size_t t) {
        for (size_t u = t; u < units; u += nt) {
            const size_t r = u / pieces, c0 = (u % pieces) * pw;
            if (c0 >= width) continue;
            const size_t w = width - c0 < pw ? width - c0 : pw;
            std::memcpy(dst + r * dpitch + c0, src + r * spitch + c0, w * sizeof(double));
        }
    };
    if (nt == 1) return work(0);
    std::vector<std::thread> th;
    for (size_t t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
}

int dalloc(apv_handle* h, double** p, size_t count) {
    BCHK(h, hipMalloc((void**)p, sizeof(double) * (count ? count : 1)));
    BCHK(h, hipMemsetAsync(*p, 0, sizeof(double) * (count ? count : 1), h->stream));
    return APV_OK;
}

using d4 = __attribute__((ext_vector_type(4))) double;

// new_hist = [old_hist[H:], x, zeros(pad)]: the last P-1 inputs, this hop, and the zero tail
__global__ void __launch_bounds__(256) hist_f64_kernel(int P, int H, int pad, const double* __restrict__ old_hist,
                                                       const double* __restrict__ x, double* __restrict__ new_hist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int keep = P - 1;
    if (i < keep) new_hist[i] = old_hist[i + H];
    else if (i < keep + H) new_hist[i] = x[i - keep];
    else if (i < keep + H + pad) new_hist[i] = 0.0;
}

// both input signals' histories in one launch (blockIdx.y = signal)
__global__ void __launch_bounds__(256) hist2_f64_kernel(int P, int H, int pad, const double* __restrict__ old0,
                                                        const double* __restrict__ old1, const double* __restrict__ x,
                                                        double* __restrict__ new0, double* __restrict__ new1) {
    const int g = blockIdx.y;
    const double* old_hist = g ? old1 : old0;
    double* new_hist = g ? new1 : new0;
    x += (size_t)g * H;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int keep = P - 1;
    if (i < keep) new_hist[i] = old_hist[i + H];
    else if (i < keep + H) new_hist[i] = x[i - keep];
    else if (i < keep + H + pad) new_hist[i] = 0.0;
}

// ring[c][(len - H + n + off) % len] = src[c][n], n < H   (src row stride src_ld)
__global__ void __launch_bounds__(256) ring_append_f64_kernel(int len, int H, int off, const double* __restrict__ src,
                                                              long src_ld, double* __restrict__ ring) {
    const int c = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n < H) ring[(size_t)c * len + (len - H + n + off) % len] = src[(size_t)c * src_ld + n];
}

// ---- statistics -----------------------------------------------------------------------------------
// logical sample t of the "toeplitz" sequence g = buf with sample J removed (apvast.py:336-338)
// (skip = 0: plain Hankel data matrix, used by the static solver)
__device__ __forceinline__ double stat_at(const double* __restrict__ ring, int S, int off, int J, int t, int skip = 1) {
    const int u = (t < J || !skip) ? t : t + 1;
    int ph = u + off;
    if (ph >= S) ph -= S;
    return ring[ph];
}


// R[rho][sigma] = sum_m sum_ncol Y_m[rho][ncol] Y_m[sigma][ncol], Y_m[(s, i)][ncol] = g_{s,m}[J-1-i+ncol].
//
// A workgroup of four waves owns a 32 x 32 tile of R (one 16 x 16 v_mfma_f64_16x16x4_f64 accumulator per wave).
// A row (s, i) of Y is a window of the sequence g_{s,m}, so a tile side only ever needs, per microphone and per
// chunk of TC columns, one window of TC + 31 samples for each loudspeaker its 32 rows touch: those windows are
// unwrapped from the rings into LDS once (double-buffered: the next set is fetched into registers while the MFMAs
// of the current one run) and every operand is then an LDS read at (window base + column).
constexpr int SY_PER = 10;                      // window doubles staged per thread and step
constexpr int SY_BUF = 256 * SY_PER;            // doubles per LDS buffer (both sides, all segments)
struct SyrkJobs {
    const double* stats[4];
    double* R[4];
};

__global__ void __launch_bounds__(256) syrk_hankel_kernel(int n, int J, int L, int M, int S, int off, int skip, int ncols,
                                                          int TC, int nseg, SyrkJobs jobs) {
    extern __shared__ double sy_lds[];           // [2][SY_BUF]
    const double* __restrict__ stats = jobs.stats[blockIdx.z];
    double* __restrict__ R = jobs.R[blockIdx.z];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int W = TC + 32;
    const int nchunks = (ncols + TC - 1) / TC, iters = M * nchunks;
    const int tmax = skip ? S - 2 : S - 1;
    const int slo[2] = {r0 / J, c0 / J};
    const int t0[2] = {r0, c0};
    // this lane's rows on the two sides of its wave tile
    const int ra = r0 + (wave >> 1) * 16 + (lane & 15), cb = c0 + (wave & 1) * 16 + (lane & 15);
    int offA = -1, offB = -1;
    if (ra < n) {
        const int sa = ra / J, ia = ra - sa * J;
        const int ihi = min(r0 + 31, (sa + 1) * J - 1) - sa * J;
        offA = (sa - slo[0]) * W + (ihi - ia) + kq;
    }
    if (cb < n) {
        const int sb = cb / J, ib = cb - sb * J;
        const int ihi = min(c0 + 31, (sb + 1) * J - 1) - sb * J;
        offB = (nseg + sb - slo[1]) * W + (ihi - ib) + kq;
    }
    // staging: element e of the buffer = (side, segment, w)
    const int total = 2 * nseg * W;
    double stage[SY_PER];
    auto fetch = [&](int it) {
        const int m = it / nchunks, nc0 = (it - m * nchunks) * TC;
#pragma unroll
        for (int q = 0; q < SY_PER; ++q) {
            const int e = tid + q * 256;
            double v = 0.0;
            if (e < total) {
                const int sg = e / W, w = e - sg * W;
                const int side = sg >= nseg, sp = slo[side] + (side ? sg - nseg : sg);
                if (sp < L) {
                    const int ihi = min(t0[side] + 31, (sp + 1) * J - 1) - sp * J;
                    const int t = nc0 + J - 1 - ihi + w;
                    if (t <= tmax && ihi >= 0) v = stat_at(stats + (size_t)(m * L + sp) * S, S, off, J, t, skip);
                }
            }
            stage[q] = v;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int q = 0; q < SY_PER; ++q) {
            const int e = tid + q * 256;
            if (e < total) sy_lds[buf * SY_BUF + e] = stage[q];
        }
    };
    d4 acc = {0, 0, 0, 0};
    fetch(0);
    commit(0);
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
        if (it + 1 < iters) fetch(it + 1);
        const int nc0 = (it % nchunks) * TC;
        const int steps = (min(TC, ncols - nc0) + 3) >> 2;
        const double* la = sy_lds + buf * SY_BUF + (offA >= 0 ? offA : 0);
        const double* lb = sy_lds + buf * SY_BUF + (offB >= 0 ? offB : 0);
        const bool tail = nc0 + 4 * steps > ncols;          // the last step of the last chunk may run past the columns
        const int full_steps = tail ? steps - 1 : steps;
        int k = 0;
        for (; k + 4 <= full_steps; k += 4) {
            const double a0 = la[4 * k], a1 = la[4 * k + 4], a2 = la[4 * k + 8], a3 = la[4 * k + 12];
            const double b0 = lb[4 * k], b1 = lb[4 * k + 4], b2 = lb[4 * k + 8], b3 = lb[4 * k + 12];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(offA >= 0 ? a0 : 0.0, offB >= 0 ? b0 : 0.0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(offA >= 0 ? a1 : 0.0, offB >= 0 ? b1 : 0.0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(offA >= 0 ? a2 : 0.0, offB >= 0 ? b2 : 0.0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(offA >= 0 ? a3 : 0.0, offB >= 0 ? b3 : 0.0, acc, 0, 0, 0);
        }
        for (; k < steps; ++k) {
            const bool ok = nc0 + 4 * k + kq < ncols;
            const double a = (ok && offA >= 0) ? la[4 * k] : 0.0;
            const double b = (ok && offB >= 0) ? lb[4 * k] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
        if (it + 1 < iters) commit(buf ^ 1);
        __syncthreads();
    }
    // f64 accumulator: row = (lane>>4) + 4 t, col = lane & 15
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = r0 + (wave >> 1) * 16 + kq + 4 * t, col = c0 + (wave & 1) * 16 + (lane & 15);
        if (row < n && col < n) R[(size_t)row * n + col] = acc[t];
    }
}

// ---- the same statistics from the displacement structure of the data matrix (round 4) ---------------------------------------
// Row (s, i) of Y_m is the window of g_{s,m} that starts at u = J - 1 - i, so R[(s, i), (s', i')] = E(u, u') =
// sum_m sum_{n < ncols} g_{s,m}[u + n] g_{s',m}[u' + n] depends on the lag d = u' - u and, weakly, on where the window sits:
// E(u + 1, u' + 1) = E(u, u') - sum_m g_s[u] g_s'[u'] + sum_m g_s[u + ncols] g_s'[u' + ncols].  One full sum per lag (2 J - 1 of
// them per pair of loudspeakers) and a walk down each diagonal replace J^2 full sums: (2 J - 1)(ncols + 2 J) instead of J^2 ncols
// multiply-adds per pair and microphone -- 16 x fewer at J = 32, 41 x at J = 100 (apvast.py:334-347 forms Y and Y Y^T outright;
// numpy's own summation order differs from any of these by the same few ulp).  A workgroup owns a pair (s <= s') of one
// matrix: the unwrapped sequences of all microphones of both loudspeakers sit in LDS (2 M (S + 8) doubles: 128 KB at the
// reference's parameters); the full sums take four consecutive lags per thread (one new LDS read per lag block and sample)
// and split the column range over the thread groups; the walks take one lag per thread.  R's mirror image is written too.
constexpr int HC_TPB = 1024, HC_MAXPARTS = 10, HC_MAXSEG = 16;
// doubles of the work array behind the sequences: the full sums' partials [parts][lag blocks][4], later the walks' segment sums
__host__ __device__ inline int hc_red_len(int J) {
    const int nbk = (J + 3) / 4 + (J - 1 + 3) / 4, nqmax = 2 * J - 1;
    int nseg = HC_TPB / nqmax;
    nseg = nseg < 1 ? 1 : (nseg > HC_MAXSEG ? HC_MAXSEG : nseg);
    const int a = HC_MAXPARTS * nbk * 4, b = nqmax * nseg;
    return a > b ? a : b;
}
__global__ void __launch_bounds__(HC_TPB) hankel_corr_kernel(int n, int J, int L, int M, int S, int off, int skip, int ncols, int Sg,
                                                             SyrkJobs jobs) {
    extern __shared__ double hc_lds[];           // g[2][M][Sg], then red[parts][NBK][4] (later: seg[nq][nseg]), then e0[2 J]
    const double* __restrict__ stats = jobs.stats[blockIdx.z];
    double* __restrict__ R = jobs.R[blockIdx.z];
    const int tid = threadIdx.x;
    int sa = 0, sb = blockIdx.x;                  // pair index -> s <= s'
    while (sb >= L - sa) { sb -= L - sa; ++sa; }
    sb += sa;
    const bool same = sa == sb;
    const int tmax = skip ? S - 2 : S - 1;
    double* const gA = hc_lds;
    double* const gB = hc_lds + (size_t)M * Sg;
    for (int e = tid; e < 2 * M * Sg; e += HC_TPB) {
        const int side = e / (M * Sg), r = e - side * M * Sg, m = r / Sg, t = r - m * Sg;
        const int sp = side ? sb : sa;
        hc_lds[e] = t <= tmax ? stat_at(stats + (size_t)(m * L + sp) * S, S, off, J, t, skip) : 0.0;
    }
    __syncthreads();
    // lag jobs: q < J: d = q >= 0 (A = g_s, B = g_s', u = 0, u' = d); q >= J: d = -(q - J + 1) (A = g_s', B = g_s, u' = 0, u = -d)
    const int nq = same ? J : 2 * J - 1;
    const int nb_pos = (J + 3) / 4, nb_neg = same ? 0 : (J - 1 + 3) / 4, NBK = nb_pos + nb_neg;
    int parts = HC_TPB / NBK;
    parts = parts < 1 ? 1 : (parts > HC_MAXPARTS ? HC_MAXPARTS : parts);
    double* const red = hc_lds + (size_t)2 * M * Sg;                 // [parts][NBK][4]; the walks reuse it as seg[nq][nseg]
    const int red_len = hc_red_len(J);
    double* const e0 = red + red_len;                                 // [2 J]
    for (int item = tid; item < NBK * parts; item += HC_TPB) {
        // neighbouring lanes take neighbouring column ranges of the same lag block, an ODD number of samples apart: their LDS reads
        // fall into different banks (lanes on neighbouring lag blocks read 32 bytes apart: a four-way conflict on every access)
        const int part = item % parts, blk = item / parts;
        const bool neg = blk >= nb_pos;
        const int dd0 = neg ? 1 + 4 * (blk - nb_pos) : 4 * blk;
        const double* const A = neg ? gB : gA;
        const double* const B = neg ? gA : gB;
        const int per = ((ncols + parts - 1) / parts) | 1, n0 = min(ncols, part * per), n1 = min(ncols, n0 + per);
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int m = 0; m < M; ++m) {
            const double* const Am = A + (size_t)m * Sg;
            const double* const Bm = B + (size_t)m * Sg + dd0;
            if (n0 < n1) {
                double b0 = Bm[n0], b1 = Bm[n0 + 1], b2 = Bm[n0 + 2];
#pragma unroll 4
                for (int nn = n0; nn < n1; ++nn) {
                    const double b3 = Bm[nn + 3], a = Am[nn];
                    a0 = __builtin_fma(a, b0, a0);
                    a1 = __builtin_fma(a, b1, a1);
                    a2 = __builtin_fma(a, b2, a2);
                    a3 = __builtin_fma(a, b3, a3);
                    b0 = b1; b1 = b2; b2 = b3;
                }
            }
        }
        double* o = red + ((size_t)part * NBK + blk) * 4;
        o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
    }
    __syncthreads();
    for (int q = tid; q < nq; q += HC_TPB) {
        const bool neg = q >= J;
        const int dd = neg ? q - J + 1 : q;
        const int blk = neg ? nb_pos + (dd - 1) / 4 : dd / 4, sub = neg ? (dd - 1) % 4 : dd % 4;
        double v = 0.0;
        for (int part = 0; part < parts; ++part) v += red[((size_t)part * NBK + blk) * 4 + sub];
        e0[q] = v;
    }
    __syncthreads();
    // The walks, E(u + 1, u' + 1) = E(u, u') + delta(u), cut into nseg segments per diagonal so that every thread has one: first
    // every segment's sum of deltas, then each segment starts from e0 plus the sums of the segments before it (fixed order) and
    // walks, forming its deltas a second time (twice the multiply-adds, nseg times the threads).
    int nseg = HC_TPB / nq;
    nseg = nseg < 1 ? 1 : (nseg > HC_MAXSEG ? HC_MAXSEG : nseg);
    const int seglen = ((J + nseg - 1) / nseg) | 1;            // odd: the segments of a diagonal start in different LDS banks
    auto delta = [&](const double* A, const double* B, int dd, int u) {
        double dl = 0.0;
        for (int m = 0; m < M; ++m) {
            const double* const Am = A + (size_t)m * Sg;
            const double* const Bm = B + (size_t)m * Sg;
            dl = __builtin_fma(Am[u + ncols], Bm[u + dd + ncols], dl);
            dl = __builtin_fma(-Am[u], Bm[u + dd], dl);
        }
        return dl;
    };
    double* const seg = red;
    for (int item = tid; item < nq * nseg; item += HC_TPB) {
        const int q = item / nseg, g = item - q * nseg;
        const bool neg = q >= J;
        const int dd = neg ? q - J + 1 : q, len = J - dd;
        const double* const A = neg ? gB : gA;
        const double* const B = neg ? gA : gB;
        double sum = 0.0;
        for (int u = g * seglen; u < (g + 1) * seglen && u + 1 < len; ++u) sum += delta(A, B, dd, u);
        seg[(size_t)q * nseg + g] = sum;
    }
    __syncthreads();
    for (int item = tid; item < nq * nseg; item += HC_TPB) {
        const int q = item / nseg, g = item - q * nseg;
        const bool neg = q >= J;
        const int dd = neg ? q - J + 1 : q, len = J - dd;
        const double* const A = neg ? gB : gA;
        const double* const B = neg ? gA : gB;
        double E = e0[q];
        for (int gp = 0; gp < g; ++gp) E += seg[(size_t)q * nseg + gp];
        for (int u = g * seglen; u < (g + 1) * seglen && u < len; ++u) {
            // A's window starts at u, B's at u + dd: rows i = J - 1 - start
            const int iA = J - 1 - u, iB = J - 1 - (u + dd);
            const int row = (neg ? sb : sa) * J + iA, col = (neg ? sa : sb) * J + iB;
            R[(size_t)row * n + col] = E;
            R[(size_t)col * n + row] = E;
            if (u + 1 < len) E += delta(A, B, dd, u);
        }
    }
}

// window length and segment count that fit the staging buffer for this (J, S)
static void syrk_plan(int J, int ncols, int* TC, int* nseg) {
    *nseg = 31 / J + 2;                                   // loudspeakers a run of 32 rows can touch
    int W = SY_BUF / (2 * *nseg);
    int tc = ((W - 32) / 4) * 4;
    const int ncols4 = (ncols + 3) / 4 * 4;
    if (tc > ncols4) tc = ncols4;
    if (tc < 4) tc = 4;
    *TC = tc;
}

static void launch_syrk(hipStream_t st, int n, int J, int L, int M, int S, int off, int skip, int ncols, int njobs,
                        const SyrkJobs& jobs) {
    // the displacement form where its LDS fits (both loudspeakers' sequences of all microphones) and a thread has a lag block;
    // APV_SYRK_MFMA=1: the dense products on the matrix cores, as until round 4 (A/B switch)
    static const bool dense = getenv("APV_SYRK_MFMA") != nullptr;
    {
        const int Sg = S + 8;
        const size_t lds = sizeof(double) * ((size_t)2 * M * Sg + (size_t)hc_red_len(J) + 2 * (size_t)J);
        if (!dense && 2 * J - 1 <= HC_TPB && lds <= 158 * 1024 && ncols + J + 2 <= Sg) {
            static bool once = false;
            if (!once) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hankel_corr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
                once = true;
            }
            hipLaunchKernelGGL(hankel_corr_kernel, dim3(L * (L + 1) / 2, 1, njobs), dim3(HC_TPB), lds, st, n, J, L, M, S, off, skip, ncols, Sg, jobs);
            return;
        }
    }
    int TC, nseg;
    syrk_plan(J, ncols, &TC, &nseg);
    hipLaunchKernelGGL(syrk_hankel_kernel, dim3((n + 31) / 32, (n + 31) / 32, njobs), dim3(256), sizeof(double) * 2 * SY_BUF, st, n, J,
                       L, M, S, off, skip, ncols, TC, nseg, jobs);
}

// r[rho] = sum_m sum_ncol Y_m[rho][ncol] d_m[toff + ncol]      (apvast.py:340, 356: toff = J; apVast.m:425: J - 1)
__global__ void __launch_bounds__(256) xcorr_hankel_kernel(int n, int J, int L, int M, int S, int off, int skip, int ncols,
                                                           int toff, const double* __restrict__ stats,
                                                           const double* __restrict__ tstats, double* __restrict__ r) {
    const int rho = blockIdx.x;
    const int s = rho / J, i = rho % J;
    double acc = 0.0;
    for (int idx = threadIdx.x; idx < M * ncols; idx += 256) {
        const int m = idx / ncols, nc = idx - m * ncols;
        const double y = stat_at(stats + (size_t)(m * L + s) * S, S, off, J, J - 1 - i + nc, skip);
        int ph = toff + nc + off;
        if (ph >= S) ph -= S;
        acc += y * tstats[(size_t)m * S + ph];
    }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) r[rho] = red[0];
}

// lambda_max of a symmetric positive semi-definite n x n matrix (= norm(R), apVast.m:559-566): KL Lanczos steps and
// the largest eigenvalue of the resulting tridiagonal matrix by 1024-way multisection on Sturm counts.  One workgroup
// per matrix; in the matrix-vector product a wave takes a row at a time so that the loads run along it.  (Plain power
// iteration needs thousands of steps when the leading eigenvalues cluster, which they do right after start-up.)
constexpr int KL = 96;
struct NormJobs {
    const double* m[4];
};
__global__ void __launch_bounds__(1024) norm2_lanczos_kernel(int n, NormJobs jobs, double* __restrict__ out) {
    extern __shared__ double pw[];               // v [n], vp [n], w [n], alpha [KL], beta [KL + 1], red [32], flag [1025]
    const double* __restrict__ Rm = jobs.m[blockIdx.x];
    double *v = pw, *vp = pw + n, *w = pw + 2 * n, *alpha = pw + 3 * n, *beta = alpha + KL, *red = beta + KL + 1;
    int* flag = reinterpret_cast<int*>(red + 32);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    auto block_sum = [&](double a) {             // sum over the workgroup, result in every thread
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o, 64);
        __syncthreads();
        if (lane == 0) red[wave] = a;
        __syncthreads();
        double t = 0.0;
        for (int q = 0; q < 16; ++q) t += red[q];
        return t;
    };
    double nrm0 = 0.0;
    for (int i = tid; i < n; i += 1024) {
        const double x = 1.0 + 0.37 * (double)((i * 7) % 11);
        v[i] = x;
        vp[i] = 0.0;
        nrm0 += x * x;
    }
    nrm0 = block_sum(nrm0);
    const double inv0 = 1.0 / sqrt(nrm0);
    for (int i = tid; i < n; i += 1024) v[i] *= inv0;
    if (tid == 0) beta[0] = 0.0;
    __syncthreads();
    const int kmax = n < KL ? n : KL;
    int k = 0;
    for (int j = 0; j < kmax; ++j) {
        for (int row = wave; row < n; row += 16) {
            double a = 0.0;
            for (int c = lane; c < n; c += 64) a += Rm[(size_t)row * n + c] * v[c];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o, 64);
            if (lane == 0) w[row] = a;
        }
        __syncthreads();
        double d = 0.0;
        for (int i = tid; i < n; i += 1024) d += v[i] * w[i];
        const double aj = block_sum(d);
        const double bj = beta[j];
        double nn2 = 0.0;
        for (int i = tid; i < n; i += 1024) {
            const double x = w[i] - aj * v[i] - bj * vp[i];
            w[i] = x;
            nn2 += x * x;
        }
        nn2 = block_sum(nn2);
        const double bn = sqrt(nn2);
        if (tid == 0) {
            alpha[j] = aj;
            beta[j + 1] = bn;
        }
        k = j + 1;
        if (!(bn > 1e-14 * fabs(aj)) || !(bn > 1e-300)) break;          // invariant subspace found (uniform)
        const double ib = 1.0 / bn;
        for (int i = tid; i < n; i += 1024) {
            vp[i] = v[i];
            v[i] = w[i] * ib;
        }
        __syncthreads();
    }
    __syncthreads();
    // Gershgorin bracket of the tridiagonal matrix, then multisection: count(x) = number of eigenvalues below x
    double lo = alpha[0], hi = alpha[0];
    for (int i = 0; i < k; ++i) {
        const double rad = (i > 0 ? fabs(beta[i]) : 0.0) + (i + 1 < k ? fabs(beta[i + 1]) : 0.0);
        lo = fmin(lo, alpha[i] - rad);
        hi = fmax(hi, alpha[i] + rad);
    }
    for (int round = 0; round < 7; ++round) {
        const double x = lo + (hi - lo) * (double)(tid + 1) / 1025.0;
        int cnt = 0;
        double dd = 1.0;
        for (int i = 0; i < k; ++i) {
            const double b2 = i > 0 ? beta[i] * beta[i] : 0.0;
            dd = (alpha[i] - x) - (i > 0 ? b2 / dd : 0.0);
            if (dd == 0.0) dd = -1e-300;
            cnt += dd < 0.0;
        }
        flag[tid + 1] = cnt >= k;                  // every eigenvalue lies below x
        if (tid == 0) flag[0] = 0;
        __syncthreads();
        // the largest eigenvalue sits between the last x that is not above all of them and the first that is
        const double step = (hi - lo) / 1025.0;
        double nlo = lo, nhi = hi;
        int first = 1025;
        for (int q = wave * 64 + lane; q < 1024; q += 1024) first = (flag[q + 1] && !flag[q]) ? q : first;
        // reduce `first` (exactly one boundary exists since the counts are monotone)
        red[0] = 0;
        __syncthreads();
        if (first < 1025) {
            red[0] = (double)first;
            red[1] = 1.0;
        }
        if (tid == 0 && !flag[1024]) red[1] = 0.0;
        __syncthreads();
        if (flag[1024]) {
            const int f = (int)red[0];
            nlo = lo + step * (double)f;
            nhi = lo + step * (double)(f + 1);
        } else {
            nlo = lo + step * 1024.0;              // numerically above the bracket: keep the top slice
        }
        __syncthreads();
        lo = nlo;
        hi = nhi;
    }
    if (tid == 0) out[blockIdx.x] = 0.5 * (lo + hi);
}

// R[i][i] += coef * nrm[q]
__global__ void __launch_bounds__(256) add_rel_diag_kernel(int n, double* __restrict__ R, double coef, const double* __restrict__ nrm) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) R[(size_t)i * n + i] += coef * nrm[0];
}

// out[ch][k] = in[k] * filt[ch][k]   (complex, channel-major)
__global__ void __launch_bounds__(256) apply_f64_kernel(int K, int n_ch, const double2* __restrict__ in,
                                                        const double2* __restrict__ filt, double2* __restrict__ out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int ch = blockIdx.y;
    if (k >= K || ch >= n_ch) return;
    const double2 x = in[k], f = filt[(size_t)ch * K + k];
    out[(size_t)ch * K + k] = make_double2(x.x * f.x - x.y * f.y, x.x * f.y + x.y * f.x);
}

// the same for up to four channel groups with inputs of their own, one launch (a hop's zone programs and target paths)
struct ApplyJobsD {
    const double2* in[4];
    const double2* filt[4];
    double2* out[4];
    int first[5];          // first channel of each job in the launch's channel index; first[n] = total
    int n;
};
__global__ void __launch_bounds__(256) apply_f64_jobs_kernel(int K, ApplyJobsD j) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    int ch = blockIdx.y, q = 0;
    while (q + 1 < j.n && ch >= j.first[q + 1]) ++q;
    ch -= j.first[q];
    if (k >= K) return;
    const double2 x = j.in[q][k], f = j.filt[q][(size_t)ch * K + k];
    j.out[q][(size_t)ch * K + k] = make_double2(x.x * f.x - x.y * f.y, x.x * f.y + x.y * f.x);
}

// p[t][m] = sum_s sum_q rir[q][s][m] x[t-q][s]          (Matlab/ControlMethods/predictPressure.m:12-16)
__global__ void __launch_bounds__(256) predict_pressure_kernel(int T, int L, int M, int P, const double* __restrict__ x,
                                                               const double* __restrict__ rir, double* __restrict__ out) {
    const int m = threadIdx.x % M, tl = threadIdx.x / M;
    const int tpb = 256 / M;
    const int t = blockIdx.x * tpb + tl;
    if (tl >= tpb || t >= T) return;
    double acc = 0.0;
    const int qmax = (t + 1 < P) ? t + 1 : P;
    for (int q = 0; q < qmax; ++q) {
        const double* xr = x + (size_t)(t - q) * L;
        const double* rr = rir + (size_t)q * L * M + m;
        for (int s = 0; s < L; ++s) acc = __builtin_fma(rr[(size_t)s * M], xr[s], acc);
    }
    out[(size_t)t * M + m] = acc;
}

__global__ void __launch_bounds__(256) scale_kernel(size_t count, double* __restrict__ v, double f) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) v[i] *= f;
}

inline int path_sig(int p) { return p >> 1; }       // AA, AB, BA, BB
inline int path_zone(int p) { return p & 1; }

}  // namespace

hipError_t apv_launch_norm2(int n, int count, const double* const* d_mats, double* d_out, hipStream_t s) {
    if (count < 1 || count > 4) return hipErrorInvalidValue;
    NormJobs nj{};
    for (int q = 0; q < count; ++q) nj.m[q] = d_mats[q];
    hipLaunchKernelGGL(norm2_lanczos_kernel, dim3(count), dim3(1024), sizeof(double) * (3 * (size_t)n + 2 * KL + 33 + 520), s, n, nj, d_out);
    return hipGetLastError();
}

void apv_bb_free(apv_handle* h) {
    apv_bb* s = h->bb;
    if (!s) return;
    double* bufs[] = {s->rir[0], s->rir[1], s->trir[0], s->trir[1], s->xhist[0][0], s->xhist[0][1], s->xhist[1][0],
                      s->xhist[1][1], s->xin, s->resp_all, s->ov_all, s->stats_all, s->pspec_all,
                      s->inblk, s->spec, s->R, s->r, s->U, s->lam, s->w,
                      s->fspec, s->inspec, s->outov, s->out, s->G2, s->G2T, s->Wgt[0], s->Wgt[1], s->nrm,
                      s->g_xin, s->g_RA, s->g_RB, s->g_U, s->g_lam, s->g_r, s->g_w, s->g_nrm, s->g_inspec, s->spec_out, s->g_out};
    for (double* b : bufs)
        if (b) (void)hipFree(b);
    if (s->g_pin) (void)hipHostFree(s->g_pin);
    if (s->g_pin_in) (void)hipHostFree(s->g_pin_in);
    if (s->d_ranks) (void)hipFree(s->d_ranks);
    if (s->pin_in) (void)hipHostFree(s->pin_in);
    if (s->pin_out) (void)hipHostFree(s->pin_out);
    if (s->front) (void)hipStreamDestroy(s->front);
    if (s->copy) (void)hipStreamDestroy(s->copy);
    if (s->front2) (void)hipStreamDestroy(s->front2);
    if (s->ev_ring) (void)hipEventDestroy(s->ev_ring);
    if (s->ev_stat) (void)hipEventDestroy(s->ev_stat);
    for (int i = 0; i < 2; ++i) {
        if (s->ev_front[i]) (void)hipEventDestroy(s->ev_front[i]);
        if (s->ev_back[i]) (void)hipEventDestroy(s->ev_back[i]);
        if (s->ev_out[i]) (void)hipEventDestroy(s->ev_out[i]);
    }
    delete s;
    h->bb = nullptr;
}

extern "C" {

// Ranks of the solutions the next apv_bb_init keeps (apVast.m:527-549 takes a vector of ranks); n_ranks = 0 restores
// "every rank 1..number_of_eigenvectors" (apvast.py:406-422).
int apv_bb_set_rank_list(apv_handle* h, int32_t n_ranks, const int32_t* ranks) {
    if (!h || n_ranks < 0 || (n_ranks > 0 && !ranks)) return apv_fail(h, APV_ERR_ARG, "bad rank list");
    h->bb_rank_list.assign(ranks, ranks + n_ranks);
    return APV_OK;
}

int apv_bb_init(apv_handle* h, int32_t rir_len, const double* h_rir_A, const double* h_rir_B,
                int32_t reference_index_A, int32_t reference_index_B, int32_t modeling_delay,
                int32_t filter_length, int32_t statistics_buffer_length, int32_t number_of_eigenvectors) {
    if (!h || !h_rir_A || !h_rir_B) return apv_fail(h, APV_ERR_ARG, "null argument");
    const apv_config& c = h->cfg;
    const int N = c.block_size, H = c.hop_size, J = filter_length, S = statistics_buffer_length;
    const int V = number_of_eigenvectors, L = c.n_srcs, M = c.n_mics;
    {
        std::string why;
        if (!apv_stft_size_ok(N, &why)) return apv_fail(h, APV_ERR_ARG, why);
    }
    if (N > 4096) return apv_fail(h, APV_ERR_ARG, "broadband mode: block_size <= 4096 (double-precision FFT in LDS)");
    if (H < 1 || H > N) return apv_fail(h, APV_ERR_ARG, "hop_size must be in 1..block_size");
    if (J < 1 || J > N || S <= J + 1 || H > S) return apv_fail(h, APV_ERR_ARG, "need 1 <= filter_length <= block_size and statistics_buffer_length > filter_length + 1, >= hop_size");
    const int n = J * L;
    if (n > 2048) return apv_fail(h, APV_ERR_ARG, "broadband mode: filter_length * loudspeakers <= 2048");
    if (V < 1 || V > n) return apv_fail(h, APV_ERR_ARG, "number_of_eigenvectors must be in 1..filter_length*loudspeakers");
    // ranks kept: every rank 1..V (apvast.py:406-422) unless a list was registered (apVast.m:527-549)
    std::vector<int> ranks = h->bb_rank_list;
    if (ranks.empty())
        for (int v = 1; v <= V; ++v) ranks.push_back(v);
    for (size_t i = 0; i < ranks.size(); ++i)
        if (ranks[i] < 1 || ranks[i] > n || (i && ranks[i] <= ranks[i - 1]))
            return apv_fail(h, APV_ERR_ARG, "rank list must be ascending and within 1..filter_length*loudspeakers");
    const int dialect = c.dialect;
    if (dialect != APV_DIALECT_PYTHON && dialect != APV_DIALECT_MATLAB) return apv_fail(h, APV_ERR_ARG, "unknown dialect");
    if (rir_len < 1 || modeling_delay < 0 || modeling_delay >= rir_len || modeling_delay >= J)
        return apv_fail(h, APV_ERR_ARG, "modeling_delay must be < rir_len and < filter_length");
    if (reference_index_A < 0 || reference_index_A >= L || reference_index_B < 0 || reference_index_B >= L)
        return apv_fail(h, APV_ERR_ARG, "reference index out of range");
    if (c.n_zones < 1 || c.n_zones > 3) return apv_fail(h, APV_ERR_ARG, "n_zones is a bit mask: 1 = A, 2 = B, 3 = both");
    BCHK(h, hipSetDevice(h->device));
    apv_bb_free(h);
    apv_bb* s = new apv_bb();
    std::memset(static_cast<void*>(s), 0, sizeof(apv_bb));
    h->bb = s;
    const int nsol = (int)ranks.size();
    s->N = N; s->H = H; s->K = N / 2 + 1; s->L = L; s->M = M; s->C = L * M; s->P = rir_len; s->J = J; s->S = S; s->V = nsol;
    s->n = n; s->zones = c.n_zones; s->pad = apv_fir_pad_f64();
    s->dialect = dialect;
    s->skip = dialect == APV_DIALECT_PYTHON;                 // scipy's toeplitz drops sample J (apvast.py:336-338)
    s->ncols = S - J + (s->skip ? 0 : 1);                    // apvast.py:334 / apVast.m:420
    s->toff = s->skip ? J : J - 1;                           // apvast.py:340 d[J:] / apVast.m:425 d(J:end)
    s->rel_loading = c.reg_mode == APV_REG_REL && dialect == APV_DIALECT_MATLAB;      // apVast.m:552-569, in place
    s->rel_dark_py = c.reg_mode == APV_REG_REL && dialect == APV_DIALECT_PYTHON;      // apvast.py:26-27, inside jdiag
    const int nz = ((s->zones & 1) ? 1 : 0) + ((s->zones & 2) ? 1 : 0);
    s->n_out = nz * nsol * L + 2 * L;
    s->out_group = c.out_layout == 1 ? L : 0;
    s->max_rank = ranks.back();
    s->full_valid = 1;
    BCHK(h, hipMalloc((void**)&s->d_ranks, sizeof(int) * nsol));
    BCHK(h, hipMemcpyAsync(s->d_ranks, ranks.data(), sizeof(int) * nsol, hipMemcpyHostToDevice, h->stream));
    BCHK(h, hipStreamSynchronize(h->stream));
    const int C = s->C, P = s->P, K = s->K;
    int rc;
    std::vector<double> tmp((size_t)P * C), ttmp((size_t)P * M);
    for (int z = 0; z < 2; ++z) {
        const double* src = z ? h_rir_B : h_rir_A;
        const int ref = z ? reference_index_B : reference_index_A;
        for (int p = 0; p < P; ++p)
            for (int l = 0; l < L; ++l)
                for (int m = 0; m < M; ++m) tmp[(size_t)p * C + m * L + l] = src[((size_t)p * L + l) * M + m];
        std::fill(ttmp.begin(), ttmp.end(), 0.0);
        for (int p = modeling_delay; p < P; ++p)
            for (int m = 0; m < M; ++m) ttmp[(size_t)p * M + m] = src[((size_t)(p - modeling_delay) * L + ref) * M + m];
        if ((rc = dalloc(h, &s->rir[z], (size_t)P * C))) return rc;
        if ((rc = dalloc(h, &s->trir[z], (size_t)P * M))) return rc;
        // on the handle's stream, behind the zero-fill of dalloc (a null-stream hipMemcpy is not ordered with it)
        BCHK(h, hipMemcpyAsync(s->rir[z], tmp.data(), sizeof(double) * tmp.size(), hipMemcpyHostToDevice, h->stream));
        BCHK(h, hipMemcpyAsync(s->trir[z], ttmp.data(), sizeof(double) * ttmp.size(), hipMemcpyHostToDevice, h->stream));
        BCHK(h, hipStreamSynchronize(h->stream));           // tmp / ttmp are rewritten for the next zone
    }
    const size_t hist = (size_t)P - 1 + H + s->pad;
    for (int b = 0; b < 2; ++b)
        for (int g = 0; g < 2; ++g)
            if ((rc = dalloc(h, &s->xhist[b][g], hist))) return rc;
    if ((rc = dalloc(h, &s->xin, (size_t)2 * H))) return rc;
    s->n_all = 4 * C + 2 * M;
    if ((rc = dalloc(h, &s->resp_all, (size_t)s->n_all * N))) return rc;
    if ((rc = dalloc(h, &s->ov_all, (size_t)s->n_all * N))) return rc;
    if ((rc = dalloc(h, &s->stats_all, (size_t)s->n_all * S))) return rc;
    if ((rc = dalloc(h, &s->pspec_all, (size_t)s->n_all * K * 2))) return rc;
    for (int p = 0; p < 4; ++p) {
        s->resp[p] = s->resp_all + (size_t)p * C * N;
        s->ov[p] = s->ov_all + (size_t)p * C * N;
        s->stats[p] = s->stats_all + (size_t)p * C * S;
    }
    for (int z = 0; z < 2; ++z) {
        s->tresp[z] = s->resp_all + ((size_t)4 * C + (size_t)z * M) * N;
        s->tov[z] = s->ov_all + ((size_t)4 * C + (size_t)z * M) * N;
        s->tstats[z] = s->stats_all + ((size_t)4 * C + (size_t)z * M) * S;
        s->tspec[z] = s->pspec_all + ((size_t)4 * C + (size_t)z * M) * K * 2;
    }
    const size_t spec_ch = (size_t)(C > s->n_out ? C : s->n_out);
    if ((rc = dalloc(h, &s->spec, spec_ch * K * 2))) return rc;
    if ((rc = dalloc(h, &s->inblk, (size_t)2 * N))) return rc;
    if ((rc = dalloc(h, &s->R, (size_t)4 * n * n))) return rc;
    if ((rc = dalloc(h, &s->r, (size_t)2 * n))) return rc;
    if ((rc = dalloc(h, &s->U, (size_t)2 * n * n))) return rc;
    if ((rc = dalloc(h, &s->lam, (size_t)2 * n))) return rc;
    if ((rc = dalloc(h, &s->w, (size_t)2 * nsol * n))) return rc;
    if ((rc = dalloc(h, &s->nrm, 4))) return rc;
    if ((rc = dalloc(h, &s->fspec, (size_t)s->n_out * K * 2))) return rc;
    if ((rc = dalloc(h, &s->inspec, (size_t)2 * K * 2))) return rc;
    if ((rc = dalloc(h, &s->outov, (size_t)s->n_out * N))) return rc;
    if ((rc = dalloc(h, &s->out, (size_t)s->n_out * H))) return rc;
    BCHK(h, hipHostMalloc((void**)&s->pin_in, sizeof(double) * 2 * H, hipHostMallocDefault));
    BCHK(h, hipHostMalloc((void**)&s->pin_out, sizeof(double) * (size_t)s->n_out * H, hipHostMallocDefault));
    // target filter spectra: the Python class uses the zone-A reference for both A_t and B_t (apvast.py:389-390, 418,
    // 422), the MATLAB class one reference per zone (apVast.m:597-602)
    std::vector<double> tg((size_t)2 * L * K * 2, 0.0);
    const double PI = 3.14159265358979323846;
    for (int z = 0; z < 2; ++z)
        for (int k = 0; k < K; ++k) {
            const double ph = -2.0 * PI * (double)k * (double)modeling_delay / (double)N;
            const int ref = (z == 1 && dialect == APV_DIALECT_MATLAB) ? reference_index_B : reference_index_A;
            const size_t o = (((size_t)z * L + ref) * K + k) * 2;
            tg[o] = std::cos(ph);
            tg[o + 1] = std::sin(ph);
        }
    BCHK(h, hipMemcpyAsync(s->fspec + (size_t)nz * nsol * L * K * 2, tg.data(), sizeof(double) * tg.size(), hipMemcpyHostToDevice,
                           h->stream));
    BCHK(h, apv_stft_prepare(N, 1));
    BCHK(h, hipStreamSynchronize(h->stream));
    return APV_OK;
}

// Where one hop's statistics and solution live: the handle's own arrays for apv_bb_process_block, slices of the group buffers
// for apv_bb_process_signal.
struct BbHop {
    const double* xin;     // [2][H] the hop's input samples (device)
    double* Rq[4];         // bright A->A, B->B; dark A->B, B->A (zones that do not run: unused)
    double* r[2];          // per zone
    double* nrm;           // [4] ||R_q||_2
    double* nrm_dark;      // first dark norm in the order the batch holds the matrices (relative loading of the Python dialect)
    double* inspec;        // [2][K] c128
    double* w[2];          // per zone [V][n]
};

static BbHop bb_own_hop(apv_bb* s) {
    const size_t nn = (size_t)s->n * s->n;
    BbHop q{};
    q.xin = s->xin;
    for (int i = 0; i < 4; ++i) q.Rq[i] = s->R + i * nn;
    for (int z = 0; z < 2; ++z) {
        q.r[z] = s->r + (size_t)z * s->n;
        q.w[z] = s->w + (size_t)z * s->V * s->n;
    }
    q.nrm = s->nrm;
    q.nrm_dark = s->nrm + 2 + ((s->zones & 1) ? 0 : 1);
    q.inspec = s->inspec;
    return q;
}

// stages 1-3 of a hop (and the input spectra of stage 6, which depend on the input ring as it stands now): everything in front
// of the joint diagonalisation.  t_stage: optional wall times of the stages (APV_BB_TIMING).
// st2 (optional, apv_bb_process_signal): the statistics of the hop run there, so that the next hop's K1 / WOLA chain on `st` does not
// wait for them -- the two halves of a hop are about equal at cfg1, ~40 us of small dependent launches each.  The rings order them:
// the statistics read the rings after this hop's append (ev_ring) and the next hop's append waits for them (ev_stat).
static int bb_front(apv_handle* h, apv_bb* s, const BbHop& q, double* t_stage, hipStream_t st, hipStream_t st2 = nullptr) {
    const int N = s->N, H = s->H, K = s->K, L = s->L, M = s->M, C = s->C, P = s->P, J = s->J, S = s->S, n = s->n;
    std::string why;
    int n_stage = 0;
    auto t_prev = std::chrono::steady_clock::now();
    auto stage_done = [&]() {
        if (!t_stage) return;
        (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        t_stage[n_stage++] = std::chrono::duration<double, std::milli>(now - t_prev).count();
        t_prev = now;
    };
    const int nxt = s->cur ^ 1;
    const int hist_total = P - 1 + H + s->pad;
    hipLaunchKernelGGL(hist2_f64_kernel, dim3((hist_total + 255) / 256, 2), dim3(256), 0, st, P, H, s->pad, s->xhist[s->cur][0],
                       s->xhist[s->cur][1], q.xin, s->xhist[nxt][0], s->xhist[nxt][1]);
    s->cur = nxt;
    s->ring_off = (s->ring_off + H) % N;
    s->stat_off = (s->stat_off + H) % S;
    hipLaunchKernelGGL(ring_append_f64_kernel, dim3((H + 255) / 256, 2), dim3(256), 0, st, N, H, s->ring_off, q.xin,
                       (long)H, s->inblk);
    // 1: RIR convolution
    {
        FirJobsD jobs{};
        int nj = 0;
        for (int p = 0; p < 4; ++p) {
            jobs.rir[nj] = s->rir[path_zone(p)]; jobs.xh[nj] = s->xhist[s->cur][path_sig(p)]; jobs.resp[nj] = s->resp[p];
            jobs.C[nj++] = C;
        }
        for (int z = 0; z < 2; ++z) {
            jobs.rir[nj] = s->trir[z]; jobs.xh[nj] = s->xhist[s->cur][z]; jobs.resp[nj] = s->tresp[z];
            jobs.C[nj++] = M;
        }
        BCHK(h, apv_launch_fir_jobs_f64(jobs, nj, P, H, N, s->ring_off, st));
    }
    stage_done();
    // 2: WOLA (unit weights, apvast.py:326-327, or the perceptual curves) and append the finished hop to the statistics rings:
    // all 4 C + 2 M channels of the hop in one launch per step (see resp_all)
    const bool runA = s->zones & 1, runB = s->zones & 2;
    auto pspec = [&](int p) { return s->pspec_all + (size_t)p * C * K * 2; };
    BCHK(h, apv_launch_analysis(1, N, s->n_all, s->resp_all, N, N, s->ring_off, 1, s->pspec_all, K, 1, st, &why));
    for (int p = 0; p < 4; ++p) {
        const bool live = (p == 0 || p == 1) ? runA : runB;      // A->A, A->B belong to zone program A
        if (!live) BCHK(h, hipMemsetAsync(pspec(p), 0, sizeof(double) * 2 * (size_t)C * K, st));   // apvast.py:239-255: spectra stay 0
    }
    if (s->nch > 0) {
        // curves from the unweighted target spectra of both zones (apvast.py:205), then the scaling (208-209);
        // A->A, B->A x zone A's curve; A->B, B->B x zone B's (apvast.py:258-262)
        for (int z = 0; z < 2; ++z)
            BCHK(h, apv_launch_perceptual_weights_f64(K, M, s->nch, (const double2*)s->tspec[z], s->G2, s->G2T, s->Cs, s->Ca,
                                                      s->Leff, N, s->norm_mode, s->Wgt[z], st));
        for (int z = 0; z < 2; ++z)
            BCHK(h, apv_launch_scale_spectra_cm_f64(K, M, 1, (double2*)s->tspec[z], s->Wgt[z], st));
        for (int p = 0; p < 4; ++p) {
            const bool live = (p == 0 || p == 1) ? runA : runB;
            if (live) BCHK(h, apv_launch_scale_spectra_cm_f64(K, C, L, (double2*)pspec(p), s->Wgt[path_zone(p)], st));
        }
    }
    BCHK(h, apv_launch_synthesis(1, N, H, s->n_all, s->pspec_all, K, 1, s->ov_all, nullptr, st, &why));
    if (st2) BCHK(h, hipStreamWaitEvent(st, s->ev_stat, 0));          // the previous hop's statistics have read the rings
    hipLaunchKernelGGL(ring_append_f64_kernel, dim3((H + 255) / 256, s->n_all), dim3(256), 0, st, S, H, s->stat_off, s->ov_all, (long)N,
                       s->stats_all);
    hipStream_t const chain = st;
    if (st2) {
        BCHK(h, hipEventRecord(s->ev_ring, st));
        BCHK(h, hipStreamWaitEvent(st2, s->ev_ring, 0));
        st = st2;                                                      // everything up to 6a below is the statistics stage
    }
    stage_done();
    // 3: statistics.  R order: bright [0] A->A, [1] B->B; dark [2] A->B, [3] B->A
    const int stat_src[4] = {0, 3, 1, 2};
    {
        SyrkJobs jobs{};
        int nj = 0;
        for (int i = 0; i < 4; ++i) {
            const bool live = (i == 0 || i == 2) ? runA : runB;
            if (!live) continue;
            jobs.stats[nj] = s->stats[stat_src[i]];
            jobs.R[nj++] = q.Rq[i];
        }
        launch_syrk(st, n, J, L, M, S, s->stat_off, s->skip, s->ncols, nj, jobs);
    }
    if (runA) hipLaunchKernelGGL(xcorr_hankel_kernel, dim3(n), dim3(256), 0, st, n, J, L, M, S, s->stat_off, s->skip, s->ncols, s->toff, s->stats[0], s->tstats[0], q.r[0]);
    if (runB) hipLaunchKernelGGL(xcorr_hankel_kernel, dim3(n), dim3(256), 0, st, n, J, L, M, S, s->stat_off, s->skip, s->ncols, s->toff, s->stats[3], s->tstats[1], q.r[1]);
    const size_t nn = (size_t)n * n;
    if (s->dialect == APV_DIALECT_MATLAB) {
        // apVast.m:448-456: everything / ((S - J + 1) M)
        const double f = 1.0 / ((double)s->ncols * (double)M);
        for (int i = 0; i < 4; ++i) {
            const bool live = (i == 0 || i == 2) ? runA : runB;
            if (live) hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, nn, q.Rq[i], f);
        }
        for (int z = 0; z < 2; ++z)
            if (z ? runB : runA) hipLaunchKernelGGL(scale_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (size_t)n, q.r[z], f);
    }
    if (s->rel_loading || s->rel_dark_py) {
        const double* mats[4];
        for (int i = 0; i < 4; ++i) mats[i] = ((i == 0 || i == 2) ? runA : runB) ? q.Rq[i] : q.Rq[(i & 2) | (runA ? 0 : 1)];
        BCHK(h, apv_launch_norm2(n, 4, mats, q.nrm, st));
        if (q.nrm_dark != q.nrm + 2 + (runA ? 0 : 1)) {
            // the batch wants the dark norms of the zones that run side by side
            const int first = runA ? 0 : 1, cnt = (runA && runB) ? 2 : 1;
            BCHK(h, hipMemcpyAsync(q.nrm_dark, q.nrm + 2 + first, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
        }
    }
    if (s->rel_loading) {
        // apVast.m:552-569: bright += reg_bright ||R||_2, dark += reg_dark ||R||_2, in place like the reference
        for (int i = 0; i < 4; ++i) {
            const bool live = (i == 0 || i == 2) ? runA : runB;
            if (!live) continue;
            const double coef = i < 2 ? h->cfg.reg_bright : h->cfg.reg_dark;
            hipLaunchKernelGGL(add_rel_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, q.Rq[i], coef, q.nrm + i);
        }
    }
    if (st2) {
        BCHK(h, hipEventRecord(s->ev_stat, st2));
        st = chain;
    }
    // 6a: the spectra of the two input blocks as the ring holds them now
    BCHK(h, apv_launch_analysis(1, N, 2, s->inblk, N, N, s->ring_off, 1, q.inspec, K, 1, st, &why));
    stage_done();
    return APV_OK;
}

// stages 5-6 of a hop: filter spectra, outputs, overlap-add.  The hop is emitted at d_out: channel-major [n_out][H], or with
// cfg.out_layout = 1 sample-major [n_out / L][H][L] (a group = one zone program and rank, or a target path: the (hop, loudspeaker)
// array the reference's caller receives, apvast.py:498-504) with `gstride` elements between groups (0: H L).  h_out (may be null):
// one contiguous hop copied to the host, queued on the handle's stream.
static int bb_back(apv_handle* h, apv_bb* s, const BbHop& q, double* d_out, long gstride, double* h_out, double* spec) {
    hipStream_t st = h->stream;
    const int N = s->N, H = s->H, K = s->K, L = s->L, J = s->J, V = s->V;
    std::string why;
    const bool runA = s->zones & 1, runB = s->zones & 2;
    // 5: filter spectra: channel (v, l) = taps w[v][l*J : (l+1)*J] zero-padded to N, no window; both zones' filters in one launch
    // when they lie one behind the other (they do: [zone][V][n] in the handle's arrays and in a group's slices)
    const int n = s->n;
    if (runA && runB && q.w[1] == q.w[0] + (size_t)V * n) {
        BCHK(h, apv_launch_analysis(1, N, 2 * V * L, q.w[0], J, J, 0, 0, s->fspec, K, 1, st, &why));
    } else {
        int oc = 0;
        for (int z = 0; z < 2; ++z) {
            if (!(z ? runB : runA)) continue;
            BCHK(h, apv_launch_analysis(1, N, V * L, q.w[z], J, J, 0, 0, s->fspec + (size_t)oc * K * 2, K, 1, st, &why));
            oc += V * L;
        }
    }
    // 6: outputs: the live zone programs' V L channels each, then the target paths A_t, B_t, in one launch
    {
        ApplyJobsD aj{};
        int oc = 0;
        for (int z = 0; z < 2; ++z) {
            if (!(z ? runB : runA)) continue;
            aj.in[aj.n] = (const double2*)q.inspec + (size_t)z * K;
            aj.filt[aj.n] = (const double2*)s->fspec + (size_t)oc * K;
            aj.out[aj.n] = (double2*)spec + (size_t)oc * K;
            aj.first[aj.n++] = oc;
            oc += V * L;
        }
        for (int z = 0; z < 2; ++z) {
            aj.in[aj.n] = (const double2*)q.inspec + (size_t)z * K;
            aj.filt[aj.n] = (const double2*)s->fspec + (size_t)oc * K;
            aj.out[aj.n] = (double2*)spec + (size_t)oc * K;
            aj.first[aj.n++] = oc;
            oc += L;
        }
        aj.first[aj.n] = oc;
        hipLaunchKernelGGL(apply_f64_jobs_kernel, dim3((K + 255) / 256, oc), dim3(256), 0, st, K, aj);
    }
    BCHK(h, apv_launch_synthesis(1, N, H, s->n_out, spec, K, 1, s->outov, d_out, st, &why, s->out_group, gstride));
    if (h_out) BCHK(h, hipMemcpyAsync(h_out, d_out, sizeof(double) * (size_t)s->n_out * H, hipMemcpyDeviceToHost, st));
    return APV_OK;
}

int apv_bb_process_block(apv_handle* h, const double* h_in_A, const double* h_in_B, double* h_out) {
    if (!h || !h_in_A || !h_in_B || !h_out) return apv_fail(h, APV_ERR_ARG, "null argument");
    apv_bb* s = h->bb;
    if (!s) return apv_fail(h, APV_ERR_ARG, "apv_bb_init has not been called");
    BCHK(h, hipSetDevice(h->device));
    hipStream_t st = h->stream;
    const int H = s->H, V = s->V, n = s->n;
    // APV_BB_TIMING=1: synchronise at the stage boundaries and print the wall time of each (profiling aid;
    // rocprofv3 cannot trace this path)
    static const bool timing = getenv("APV_BB_TIMING") != nullptr;
    double t_stage[8] = {0};
    std::memcpy(s->pin_in, h_in_A, sizeof(double) * H);
    std::memcpy(s->pin_in + H, h_in_B, sizeof(double) * H);
    BCHK(h, hipMemcpyAsync(s->xin, s->pin_in, sizeof(double) * 2 * H, hipMemcpyHostToDevice, st));
    const BbHop q = bb_own_hop(s);
    int rc = bb_front(h, s, q, timing ? t_stage : nullptr, st);
    if (rc != APV_OK) return rc;
    const bool runA = s->zones & 1, runB = s->zones & 2;
    const size_t nn = (size_t)n * n;
    auto t_prev = std::chrono::steady_clock::now();
    // 4: jdiag + filters; both zone programs in one batch when both run
    int32_t status[2] = {0, 0};
    {
        const int first = runA ? 0 : 1, batch = (runA && runB) ? 2 : 1;
        // Python dialect, EXPERIMENTAL_REGULARIZATION = False: jdiag loads a copy of the dark matrix with reg_dark ||B||_2
        // (apvast.py:26-27); the R_* attributes stay as accumulated
        h->gl_lead_rank = s->max_rank;
        rc = apv_gevd_large(h, n, batch, s->R + first * nn, s->R + (2 + first) * nn, s->rel_loading ? 0.0 : h->cfg.reg_dark,
                            s->rel_dark_py ? s->nrm + 2 + first : nullptr, s->U + first * nn, s->lam + (size_t)first * n, s->r + (size_t)first * n, h->cfg.mu, V, s->d_ranks,
                            s->w + (size_t)first * V * n, status);
        if (rc != APV_OK) return rc;
        s->full_valid = !h->gl_lead_done;
    }
    if (timing) {
        (void)hipStreamSynchronize(st);
        t_stage[3] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_prev).count();
        t_prev = std::chrono::steady_clock::now();
    }
    // a result array from apv_host_alloc receives the hop by DMA; any other goes through the handle's page-locked buffer and a copy
    const size_t out_bytes = sizeof(double) * (size_t)s->n_out * H;
    const bool direct = host_is_pinned(h_out, out_bytes, false);
    rc = bb_back(h, s, q, s->out, 0, direct ? h_out : s->pin_out, s->spec);
    if (rc != APV_OK) return rc;
    BCHK(h, hipStreamSynchronize(st));
    BCHK(h, hipGetLastError());
    if (!direct) std::memcpy(h_out, s->pin_out, out_bytes);
    if (timing)
        fprintf(stderr, "[apv bb] fir %.3f  wola %.3f  stats %.3f  gevd %.3f  out %.3f ms\n", t_stage[0], t_stage[1],
                t_stage[2], t_stage[3], std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_prev).count());
    if (status[0] == 2 || status[1] == 2) {
        s->not_converged++;
        return apv_fail(h, APV_ERR_NO_CONVERGE, "eigen-iteration did not converge (Jacobi sweep cap reached); the outputs of this hop were written");
    }
    return APV_OK;
}

// n_hops consecutive hops in one call.  A hop's statistics depend on the input alone, never on an earlier hop's filters, so the
// joint diagonalisations of G consecutive hops are independent problems: the front stages of the G hops run one after the other
// (rings and overlap buffers are sequential state), ONE batched apv_gevd_large call solves their 2 G pairs -- a single
// n = 256 pair keeps 36 workgroups busy through ~180 dependent launches, G pairs ride in the same launches -- and the
// output stages follow in order.  Sample for sample the result is that of n_hops calls of apv_bb_process_block up to the
// rounding of the eigen-iteration (a batch sweeps until its slowest member has converged).
//                                                          replaces the hop loop of main.m:52-62 around apvast.py:153-165
int apv_bb_process_signal(apv_handle* h, int32_t n_hops, const double* h_in_A, const double* h_in_B, double* h_out) {
    if (!h || !h_in_A || !h_in_B || !h_out) return apv_fail(h, APV_ERR_ARG, "null argument");
    apv_bb* s = h->bb;
    if (!s) return apv_fail(h, APV_ERR_ARG, "apv_bb_init has not been called");
    if (n_hops < 0) return apv_fail(h, APV_ERR_ARG, "n_hops must be >= 0");
    if (n_hops == 0) return APV_OK;
    BCHK(h, hipSetDevice(h->device));
    hipStream_t st = h->stream;
    const int H = s->H, K = s->K, V = s->V, n = s->n;
    const bool runA = s->zones & 1, runB = s->zones & 2;
    const int first = runA ? 0 : 1, nz = (runA && runB) ? 2 : 1;
    const size_t nn = (size_t)n * n;
    // hops per group: eight or sixteen, or as many as keep the group's matrices (two buffer sets here, the solver's workspace: ~19 arrays of
    // n x n doubles per hop and zone) within 8 GB.  (Rounds 2-3 capped the group where the block-round launches of the batch
    // reached ~3 workgroups per compute unit, which left n = 800 at one hop per group: but a round there is its pair-solve chain,
    // not the chip -- tools/probes/large_batch_scaling.py: 7.5 / 4.9 / 4.1 ms per matrix at batch 2 / 4 / 8 -- and the whole
    // signal went 21.2 -> 15.2 / 13.0 / 11.9 ms per hop with groups of 2 / 4 / 8, under the 16.7 ms a hop of audio lasts.)
    static const int forced = getenv("APV_BB_GROUP") ? atoi(getenv("APV_BB_GROUP")) : 0;       // A/B switch
    const size_t per_hop_bytes = (size_t)19 * nn * sizeof(double) * nz;
    // sixteen while the group stays within 1 GiB (cfg1: 0.78 / 0.68 / 0.61 / 0.62 / 0.61 ms per hop with 8 / 12 / 16 / 24 / 32 hops
    // a group), eight beyond (n = 800: 11.0 ms per hop with eight, 12.0 with sixteen: 3 GB of matrices no longer sit in the
    // Infinity Cache and the batch sweeps until its slowest member is done), fewer only to stay within 8 GiB
    // (round 4: with the leading-eigenpair solver a batch no longer sweeps until its slowest member is done, and sixteen hops a
    // group are the better choice at n = 800 too: 4.83 ms per hop against 4.98 with eight, tools/probes/bb_group_sweep.sh)
    const bool lead = h->cfg.max_sweeps <= 0 && apv_gevd_lead_block(n, s->max_rank) > 0;
    int G;
    if (forced > 0) G = forced;
    else if (16 * per_hop_bytes <= ((size_t)(lead ? 8 : 1) << 30)) G = 16;
    else {
        G = (int)(((size_t)8 << 30) / per_hop_bytes);
        G = G < 1 ? 1 : (G > 8 ? 8 : G);
    }
    if (G > n_hops) G = n_hops;
    const size_t gz = (size_t)G * nz;
    // sizes of one buffer set
    const size_t z_xin = (size_t)G * 2 * H, z_mat = gz * nn, z_vec = gz * n, z_w = gz * V * n, z_nrm = (size_t)G * 4 + gz,
                 z_insp = (size_t)G * 2 * K * 2, z_out = (size_t)G * s->n_out * H;
    if (s->grp < G) {
        double** bufs[] = {&s->g_xin, &s->g_RA, &s->g_RB, &s->g_U, &s->g_lam, &s->g_r, &s->g_w, &s->g_nrm, &s->g_inspec, &s->g_out};
        for (double** b : bufs) {
            if (*b) (void)hipFree(*b);
            *b = nullptr;
        }
        s->grp = 0;
        int rc;
        if ((rc = dalloc(h, &s->g_xin, 2 * z_xin))) return rc;
        if ((rc = dalloc(h, &s->g_RA, 2 * z_mat))) return rc;
        if ((rc = dalloc(h, &s->g_RB, 2 * z_mat))) return rc;
        if ((rc = dalloc(h, &s->g_U, z_mat))) return rc;                  // U, lam: written and read by the back half only
        if ((rc = dalloc(h, &s->g_lam, z_vec))) return rc;
        if ((rc = dalloc(h, &s->g_r, 2 * z_vec))) return rc;
        if ((rc = dalloc(h, &s->g_w, z_w))) return rc;
        if ((rc = dalloc(h, &s->g_nrm, 2 * z_nrm))) return rc;
        if ((rc = dalloc(h, &s->g_inspec, 2 * z_insp))) return rc;
        if ((rc = dalloc(h, &s->g_out, 2 * z_out))) return rc;
        if (s->g_pin_in) (void)hipHostFree(s->g_pin_in);
        s->g_pin_in = nullptr;
        BCHK(h, hipHostMalloc((void**)&s->g_pin_in, sizeof(double) * 2 * z_xin, hipHostMallocDefault));
        if (!s->spec_out && (rc = dalloc(h, &s->spec_out, (size_t)s->n_out * K * 2))) return rc;
        if (!s->front) {
            BCHK(h, hipStreamCreateWithFlags(&s->front, hipStreamNonBlocking));
            // a copy stream of its own only where a group's outputs are large enough for the next group's solve to notice the
            // transfer (84 MB at the reference's test parameters; 131 KB at cfg1: there the copy rides on the handle's stream --
            // every stream fewer is one fewer to share the runtime's few hardware queues with)
            {
                static const int own = getenv("APV_BB_COPY_STREAM") ? atoi(getenv("APV_BB_COPY_STREAM")) : -1;      // A/B: 1 always, 0 never
                const size_t group_bytes = sizeof(double) * (size_t)G * s->n_out * s->H;
                if (own == 1 || (own != 0 && group_bytes >= ((size_t)4 << 20))) BCHK(h, hipStreamCreateWithFlags(&s->copy, hipStreamNonBlocking));
            }
            BCHK(h, hipStreamCreateWithFlags(&s->front2, hipStreamNonBlocking));
            BCHK(h, hipEventCreateWithFlags(&s->ev_ring, hipEventDisableTiming));
            BCHK(h, hipEventCreateWithFlags(&s->ev_stat, hipEventDisableTiming));
            for (int i = 0; i < 2; ++i) {
                BCHK(h, hipEventCreateWithFlags(&s->ev_front[i], hipEventDisableTiming));
                BCHK(h, hipEventCreateWithFlags(&s->ev_back[i], hipEventDisableTiming));
                BCHK(h, hipEventCreateWithFlags(&s->ev_out[i], hipEventDisableTiming));
            }
        }
        BCHK(h, hipStreamSynchronize(st));            // the zero fills
        s->grp = G;
    }
    // (a handle whose group buffers were sized for a larger G keeps them: the set offsets below follow the CURRENT G)
    // The caller's array: [n_out / L][n_hops H][L] with cfg.out_layout = 1 (every group's samples of the whole signal in a row: what
    // the class hands out without touching it), else [n_hops][n_out][H].  A group of hops is written on the device as the slice of
    // that array it is, and goes to the host in ONE copy: by DMA straight into the caller's array when that is page-locked
    // (apv_host_alloc), else into the staging set, from where the host moves it while the device works on the next group.
    // (Before: one copy per hop into the pageable array -- which the runtime stages and waits for -- and the class transposed the
    // whole signal afterwards: at the reference's test parameters, 5.2 MB a hop, that was 2.9 of the 3.5 ms a hop took.)
    const int og = s->out_group;
    const size_t ngrp = og > 0 ? (size_t)s->n_out / og : 1, HL = (size_t)H * (og > 0 ? og : s->n_out);
    const bool direct = host_is_pinned(h_out, sizeof(double) * (size_t)n_hops * s->n_out * H, true);
    if (!direct && s->g_pin_cap < z_out) {
        if (s->g_pin) (void)hipHostFree(s->g_pin);
        s->g_pin = nullptr;
        s->g_pin_cap = 0;
        BCHK(h, hipHostMalloc((void**)&s->g_pin, sizeof(double) * 2 * z_out, hipHostMallocDefault));
        s->g_pin_cap = z_out;
    }
    hipStream_t fs = s->front;
    // every exit below drains both streams first: copies into the caller's h_out may be in flight
    hipStream_t cs = s->copy ? s->copy : st;
    // the statistics of a hop on a stream of their own beside the next hop's K1 / WOLA chain (bb_front); APV_BB_FRONT2=0: one stream
    static const bool front2 = getenv("APV_BB_FRONT2") == nullptr || atoi(getenv("APV_BB_FRONT2")) != 0;
    hipStream_t fs2 = front2 ? s->front2 : nullptr;
    // The front stages of group g + 1 (~18 runtime calls a hop) are enqueued by a helper thread while this one solves group g: the
    // solve's passes block the host, and with the statistics on a stream of their own a group is bound by the host's enqueueing,
    // not by the device any more (0.6 ms of a group's 1.6 at cfg1).  (Tried earlier in the round, when a group was still bound
    // by its front stages on the device: no gain then.)  APV_BB_FRONT_THREAD=0: enqueue them from this thread, before the solve.
    static const bool front_threaded = getenv("APV_BB_FRONT_THREAD") == nullptr || atoi(getenv("APV_BB_FRONT_THREAD")) != 0;
    std::thread front_thr;
    int front_rc = APV_OK;
    auto drained = [&](int rc) {
        if (front_thr.joinable()) front_thr.join();
        (void)hipStreamSynchronize(fs); (void)hipStreamSynchronize(s->front2); (void)hipStreamSynchronize(st); (void)hipStreamSynchronize(cs);
        return rc;
    };
#define BDCHK(h, call)                                                                                                     \
    do {                                                                                                                    \
        hipError_t _e = (call);                                                                                             \
        if (_e != hipSuccess) return drained(apv_fail(h, APV_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e))); \
    } while (0)
    const int n_groups = (n_hops + G - 1) / G;
    std::vector<BbHop> hops[2];
    // the front stages of group g, all on the front stream (rings, histories and overlap buffers are sequential state)
    auto enqueue_front = [&](int g) -> int {
        auto drained = [](int rc) { return rc; };          // (may run on the helper thread: the caller drains)
        const int set = g & 1, h0 = g * G, g_n = n_hops - h0 < G ? n_hops - h0 : G;
        double* xin = s->g_xin + set * z_xin;
        // (one copy per group from page-locked staging: two copies per hop from the caller's pageable arrays were 32 blocking calls)
        double* const pin = s->g_pin_in + set * z_xin;
        if (g >= 2) (void)hipEventSynchronize(s->ev_front[set]);        // group g - 2's copy out of this staging set has run
        for (int i = 0; i < g_n; ++i) {
            std::memcpy(pin + (size_t)i * 2 * H, h_in_A + (size_t)(h0 + i) * H, sizeof(double) * H);
            std::memcpy(pin + (size_t)i * 2 * H + H, h_in_B + (size_t)(h0 + i) * H, sizeof(double) * H);
        }
        BDCHK(h, hipMemcpyAsync(xin, pin, sizeof(double) * (size_t)g_n * 2 * H, hipMemcpyHostToDevice, fs));
        hops[set].assign(g_n, BbHop{});
        for (int i = 0; i < g_n; ++i) {
            BbHop& q = hops[set][i];
            q.xin = xin + (size_t)i * 2 * H;
            for (int z = 0; z < 2; ++z) {
                if (!(z ? runB : runA)) continue;
                const size_t slot = (size_t)i * nz + (z - first);
                q.Rq[z] = s->g_RA + set * z_mat + slot * nn;
                q.Rq[2 + z] = s->g_RB + set * z_mat + slot * nn;
                q.r[z] = s->g_r + set * z_vec + slot * n;
                q.w[z] = s->g_w + slot * V * n;
            }
            q.nrm = s->g_nrm + set * z_nrm + (size_t)i * 4;
            q.nrm_dark = s->g_nrm + set * z_nrm + (size_t)G * 4 + (size_t)i * nz;
            q.inspec = s->g_inspec + set * z_insp + (size_t)i * 2 * K * 2;
            const int rc = bb_front(h, s, q, nullptr, fs, fs2);
            if (rc != APV_OK) return drained(rc);
        }
        if (fs2) BDCHK(h, hipStreamWaitEvent(fs, s->ev_stat, 0));          // the last hop's statistics belong to the group
        BDCHK(h, hipEventRecord(s->ev_front[set], fs));
        return APV_OK;
    };
    std::vector<int32_t> status((size_t)G * nz, 0);
    long bad_hops = 0;
    // per-hop calls before this one ran on the handle's stream: the front stream starts behind them
    BDCHK(h, hipEventRecord(s->ev_back[0], st));
    BDCHK(h, hipStreamWaitEvent(fs, s->ev_back[0], 0));
    int rc = enqueue_front(0);
    if (rc != APV_OK) return drained(rc);
    static const bool timing = getenv("APV_BB_TIMING") != nullptr;
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int g = 0; g < n_groups; ++g) {
        const int set = g & 1, h0 = g * G, g_n = n_hops - h0 < G ? n_hops - h0 : G;
        const double t_a = now_ms();
        if (g + 1 < n_groups) {
            // group g + 1 fills the other set, which the back half of group g - 1 has read (its event is on the handle's stream)
            if (g >= 1) BDCHK(h, hipStreamWaitEvent(fs, s->ev_back[set ^ 1], 0));
            if (front_threaded) {
                front_rc = APV_OK;
                front_thr = std::thread([&, g] {
                    (void)hipSetDevice(h->device);
                    front_rc = enqueue_front(g + 1);
                });
            } else if ((rc = enqueue_front(g + 1)) != APV_OK) return drained(rc);
        }
        BDCHK(h, hipStreamWaitEvent(st, s->ev_front[set], 0));
        double t_b = now_ms(), t_b2 = t_b;
        if (timing) {
            (void)hipStreamSynchronize(st);         // the group's front stages alone
            t_b2 = now_ms();
        }
        h->gl_lead_rank = s->max_rank;
        rc = apv_gevd_large(h, n, g_n * nz, s->g_RA + set * z_mat, s->g_RB + set * z_mat, s->rel_loading ? 0.0 : h->cfg.reg_dark,
                            s->rel_dark_py ? s->g_nrm + set * z_nrm + (size_t)G * 4 : nullptr, s->g_U, s->g_lam, s->g_r + set * z_vec,
                            h->cfg.mu, V, s->d_ranks, s->g_w, status.data());
        if (rc != APV_OK) return drained(rc);
        if (g + 1 == n_groups) s->full_valid = !h->gl_lead_done;
        const double t_c = now_ms();
        double* const gout = s->g_out + set * z_out;
        if (g >= 2) BDCHK(h, hipStreamWaitEvent(st, s->ev_out[set], 0));          // group g - 2 has left this output set
        for (int i = 0; i < g_n; ++i) {
            rc = og > 0 ? bb_back(h, s, hops[set][i], gout + (size_t)i * HL, (long)((size_t)G * HL), nullptr, s->spec_out)
                        : bb_back(h, s, hops[set][i], gout + (size_t)i * HL, 0, nullptr, s->spec_out);
            if (rc != APV_OK) return drained(rc);
            bool bad = false;
            for (int z = 0; z < nz; ++z) bad = bad || status[(size_t)i * nz + z] == 2;
            bad_hops += bad;
        }
        // the attributes of the handle are those of the last hop (apvast.py:368-403)
        if (g + 1 == n_groups) {
            const BbHop& q = hops[set][g_n - 1];
            for (int z = 0; z < 2; ++z) {
                if (!(z ? runB : runA)) continue;
                const size_t slot = (size_t)(g_n - 1) * nz + (z - first);
                BDCHK(h, hipMemcpyAsync(s->R + (size_t)z * nn, q.Rq[z], sizeof(double) * nn, hipMemcpyDeviceToDevice, st));
                BDCHK(h, hipMemcpyAsync(s->R + (size_t)(2 + z) * nn, q.Rq[2 + z], sizeof(double) * nn, hipMemcpyDeviceToDevice, st));
                BDCHK(h, hipMemcpyAsync(s->r + (size_t)z * n, q.r[z], sizeof(double) * n, hipMemcpyDeviceToDevice, st));
                BDCHK(h, hipMemcpyAsync(s->w + (size_t)z * V * n, q.w[z], sizeof(double) * V * n, hipMemcpyDeviceToDevice, st));
                BDCHK(h, hipMemcpyAsync(s->U + (size_t)z * nn, s->g_U + slot * nn, sizeof(double) * nn, hipMemcpyDeviceToDevice, st));
                BDCHK(h, hipMemcpyAsync(s->lam + (size_t)z * n, s->g_lam + slot * n, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
            }
            BDCHK(h, hipMemcpyAsync(s->nrm, q.nrm, sizeof(double) * 4, hipMemcpyDeviceToDevice, st));
            BDCHK(h, hipMemcpyAsync(s->inspec, q.inspec, sizeof(double) * 2 * K * 2, hipMemcpyDeviceToDevice, st));
        }
        BDCHK(h, hipEventRecord(s->ev_back[set], st));
        // the group's outputs: rows = groups of channels (one row when channel-major), g_n hops wide; on the copy stream, so that
        // the next group's solve does not wait for the transfer (84 MB = 1.7 ms of a group's 10 at the reference's test parameters)
        const size_t width = (size_t)g_n * HL, spitch = (size_t)G * HL, dpitch = (size_t)n_hops * HL;
        BDCHK(h, hipStreamWaitEvent(cs, s->ev_back[set], 0));
        if (direct) {
            BDCHK(h, hipMemcpy2DAsync(h_out + (size_t)h0 * HL, sizeof(double) * dpitch, gout, sizeof(double) * spitch,
                                      sizeof(double) * width, ngrp, hipMemcpyDeviceToHost, cs));
        } else {
            BDCHK(h, hipMemcpyAsync(s->g_pin + set * s->g_pin_cap, gout, sizeof(double) * ((ngrp - 1) * spitch + width), hipMemcpyDeviceToHost, cs));
        }
        BDCHK(h, hipEventRecord(s->ev_out[set], cs));
        // the previous group has long arrived in its staging set: move it while the device works on this one
        auto collect = [&](int gg) {
            const int cset = gg & 1, c0 = gg * G, c_n = n_hops - c0 < G ? n_hops - c0 : G;
            (void)hipEventSynchronize(s->ev_out[cset]);
            copy_rows(h_out + (size_t)c0 * HL, dpitch, s->g_pin + cset * s->g_pin_cap, spitch, (size_t)c_n * HL, ngrp);
        };
        const double t_d = now_ms();
        if (!direct && g >= 1) collect(g - 1);
        if (!direct && g + 1 == n_groups) collect(g);
        if (front_thr.joinable()) {
            front_thr.join();
            if (front_rc != APV_OK) return drained(front_rc);
        }
        if (timing) {
            const double t_e = now_ms();
            (void)hipStreamSynchronize(st);
            fprintf(stderr, "[apv bb signal] group %d (%d hops): enqueue next front %.2f, wait for this front %.2f, solve %.2f, enqueue back %.2f, "
                    "collect %.2f, drain %.2f ms\n", g, g_n, t_b - t_a, t_b2 - t_b, t_c - t_b2, t_d - t_c, t_e - t_d, now_ms() - t_e);
        }
    }
    BDCHK(h, hipStreamSynchronize(fs));
    BDCHK(h, hipStreamSynchronize(s->front2));
    BDCHK(h, hipStreamSynchronize(st));
    BDCHK(h, hipStreamSynchronize(cs));
    BDCHK(h, hipGetLastError());
    if (bad_hops) {
        s->not_converged += bad_hops;
        return apv_fail(h, APV_ERR_NO_CONVERGE, "eigen-iteration did not converge (Jacobi sweep cap reached) in some hop; the outputs were written");
    }
    return APV_OK;
}
#undef BDCHK

}  // extern "C"

long apv_bb_not_converged(const apv_handle* h) { return h->bb ? h->bb->not_converged : -1; }

extern "C" {

// Perceptual weighting for the broadband stream (see apv_stream_set_perceptual).      replaces apvast.py:313-324
int apv_bb_set_perceptual(apv_handle* h, int32_t n_channels, const double* h_G2, double Cs, double Ca, double Leff,
                          int32_t normalisation) {
    if (!h || !h->bb) return apv_fail(h, APV_ERR_ARG, "apv_bb_init has not been called");
    apv_bb* s = h->bb;
    BCHK(h, hipSetDevice(h->device));
    BCHK(h, hipStreamSynchronize(h->stream));
    if (n_channels <= 0) {
        s->nch = 0;
        return APV_OK;
    }
    if (!h_G2 || n_channels > 512 || (normalisation != 0 && normalisation != 1)) return apv_fail(h, APV_ERR_ARG, "bad perceptual tables");
    const int K = s->K;
    double* old[] = {s->G2, s->G2T, s->Wgt[0], s->Wgt[1]};
    for (double* b : old)
        if (b) (void)hipFree(b);
    s->G2 = s->G2T = nullptr;
    s->Wgt[0] = s->Wgt[1] = nullptr;
    std::vector<double> gt((size_t)n_channels * K);
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < n_channels; ++i) gt[(size_t)i * K + k] = h_G2[(size_t)k * n_channels + i];
    int rc;
    if ((rc = dalloc(h, &s->G2, (size_t)K * n_channels))) return rc;
    if ((rc = dalloc(h, &s->G2T, (size_t)K * n_channels))) return rc;
    for (int z = 0; z < 2; ++z)
        if ((rc = dalloc(h, &s->Wgt[z], (size_t)K * s->M))) return rc;
    // behind the zero-fills of dalloc, on the same (the handle's) stream
    BCHK(h, hipMemcpyAsync(s->G2, h_G2, sizeof(double) * (size_t)K * n_channels, hipMemcpyHostToDevice, h->stream));
    BCHK(h, hipMemcpyAsync(s->G2T, gt.data(), sizeof(double) * gt.size(), hipMemcpyHostToDevice, h->stream));
    BCHK(h, hipStreamSynchronize(h->stream));               // gt goes out of scope
    s->nch = n_channels; s->Cs = Cs; s->Ca = Ca; s->Leff = Leff; s->norm_mode = normalisation;
    return APV_OK;
}

// Evaluation: pressure at the microphones for given loudspeaker signals.  h_x [T][L], h_rir [P][L][M] (the
// (rir_len, L, M) layout of rirs.mat), h_out [T][M], all float64.      replaces predictPressure.m:1-17
int apv_predict_pressure(apv_handle* h, int32_t T, int32_t L, int32_t M, int32_t P, const double* h_x,
                         const double* h_rir, double* h_out) {
    if (!h || !h_x || !h_rir || !h_out) return apv_fail(h, APV_ERR_ARG, "null argument");
    if (T < 0 || L < 1 || M < 1 || M > 256 || P < 1) return apv_fail(h, APV_ERR_ARG, "predict_pressure: bad sizes (M <= 256)");
    if (T == 0) return APV_OK;
    BCHK(h, hipSetDevice(h->device));
    struct Tmp {
        double* p[3] = {nullptr, nullptr, nullptr};
        ~Tmp() { for (double* q : p) if (q) (void)hipFree(q); }
    } tmpbufs;                                              // freed on every way out
    double*& dx = tmpbufs.p[0]; double*& dr = tmpbufs.p[1]; double*& dout = tmpbufs.p[2];
    BCHK(h, hipMalloc((void**)&dx, sizeof(double) * (size_t)T * L));
    BCHK(h, hipMalloc((void**)&dr, sizeof(double) * (size_t)P * L * M));
    BCHK(h, hipMalloc((void**)&dout, sizeof(double) * (size_t)T * M));
    BCHK(h, hipMemcpyAsync(dx, h_x, sizeof(double) * (size_t)T * L, hipMemcpyHostToDevice, h->stream));
    BCHK(h, hipMemcpyAsync(dr, h_rir, sizeof(double) * (size_t)P * L * M, hipMemcpyHostToDevice, h->stream));
    const int tpb = 256 / M;
    hipLaunchKernelGGL(predict_pressure_kernel, dim3((T + tpb - 1) / tpb), dim3(256), 0, h->stream, T, L, M, P, dx, dr, dout);
    BCHK(h, hipMemcpyAsync(h_out, dout, sizeof(double) * (size_t)T * M, hipMemcpyDeviceToHost, h->stream));
    BCHK(h, hipStreamSynchronize(h->stream));
    return APV_OK;
}

// Static (signal-independent) VAST from the impulse responses alone.  h_gB [Nb][P][L], h_gD [Nd][P][L] (the
// (mics, rirLength, sources) layout of vast.m), h_w [J L].               replaces Matlab/ControlMethods/vast.m:46-91
int apv_vast_static(apv_handle* h, int32_t Nb, int32_t Nd, int32_t P, int32_t L, int32_t J, int32_t modeling_delay,
                    int32_t reference_index, int32_t V, double mu, const double* h_gB, const double* h_gD, double* h_w) {
    if (!h || !h_gB || !h_gD || !h_w) return apv_fail(h, APV_ERR_ARG, "null argument");
    const int n = J * L;
    if (Nb < 1 || Nd < 1 || P < 1 || L < 1 || J < 1 || n > 2048 || V < 1 || V > n || modeling_delay < 0 || modeling_delay >= P ||
        reference_index < 0 || reference_index >= L || P <= J)
        return apv_fail(h, APV_ERR_ARG, "vast_static: bad sizes (J L <= 2048, rir_len > J)");
    BCHK(h, hipSetDevice(h->device));
    hipStream_t st = h->stream;
    // vast.m drives the filters with a unit impulse for N = 1000 steps (vast.m:50-53): 999 of them are non-trivial
    const int ncols = 999, S = ncols + J;
    auto pack = [&](const double* g, int Nm, std::vector<double>& out) {          // [m*L + s][S], g'[u] = g[u-(J-1)]
        out.assign((size_t)Nm * L * S, 0.0);
        for (int m = 0; m < Nm; ++m)
            for (int sI = 0; sI < L; ++sI)
                for (int q = 0; q < P && q + J - 1 < S; ++q)
                    out[((size_t)m * L + sI) * S + q + J - 1] = g[((size_t)m * P + q) * L + sI];
    };
    std::vector<double> sb, sd, td((size_t)Nb * S, 0.0);
    pack(h_gB, Nb, sb);
    pack(h_gD, Nd, sd);
    for (int m = 0; m < Nb; ++m)                                                   // target: delayed reference RIR (vast.m:59)
        for (int q = modeling_delay; q < P && J + q < S; ++q)
            td[(size_t)m * S + J + q] = h_gB[((size_t)m * P + (q - modeling_delay)) * L + reference_index];
    struct Tmp {
        double* p[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        ~Tmp() { for (double* q : p) if (q) (void)hipFree(q); }
    } tmpbufs;                                              // freed on every way out
    double*& dsb = tmpbufs.p[0]; double*& dsd = tmpbufs.p[1]; double*& dtd = tmpbufs.p[2]; double*& dR = tmpbufs.p[3];
    double*& dr = tmpbufs.p[4]; double*& dU = tmpbufs.p[5]; double*& dl = tmpbufs.p[6]; double*& dw = tmpbufs.p[7];
    BCHK(h, hipMalloc((void**)&dsb, sizeof(double) * sb.size()));
    BCHK(h, hipMalloc((void**)&dsd, sizeof(double) * sd.size()));
    BCHK(h, hipMalloc((void**)&dtd, sizeof(double) * td.size()));
    BCHK(h, hipMalloc((void**)&dR, sizeof(double) * 2 * (size_t)n * n));
    BCHK(h, hipMalloc((void**)&dr, sizeof(double) * n));
    BCHK(h, hipMalloc((void**)&dU, sizeof(double) * (size_t)n * n));
    BCHK(h, hipMalloc((void**)&dl, sizeof(double) * n));
    BCHK(h, hipMalloc((void**)&dw, sizeof(double) * (size_t)V * n));
    BCHK(h, hipMemcpyAsync(dsb, sb.data(), sizeof(double) * sb.size(), hipMemcpyHostToDevice, st));
    BCHK(h, hipMemcpyAsync(dsd, sd.data(), sizeof(double) * sd.size(), hipMemcpyHostToDevice, st));
    BCHK(h, hipMemcpyAsync(dtd, td.data(), sizeof(double) * td.size(), hipMemcpyHostToDevice, st));
    const dim3 sg((n + 15) / 16, (n + 15) / 16);
    {
        SyrkJobs jb{}, jd{};
        jb.stats[0] = dsb; jb.R[0] = dR;
        jd.stats[0] = dsd; jd.R[0] = dR + (size_t)n * n;
        launch_syrk(st, n, J, L, Nb, S, 0, 0, S - J, 1, jb);
        launch_syrk(st, n, J, L, Nd, S, 0, 0, S - J, 1, jd);
    }
    hipLaunchKernelGGL(xcorr_hankel_kernel, dim3(n), dim3(256), 0, st, n, J, L, Nb, S, 0, 0, S - J, J, dsb, dtd, dr);
    // vast.m:71-73 normalises all three by numberOfMics (of the BRIGHT zone) * (rirLength - filterLength)
    const double f = 1.0 / ((double)Nb * (double)(P - J));
    const size_t nn = (size_t)n * n;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((2 * nn + 255) / 256)), dim3(256), 0, st, 2 * nn, dR, f);
    hipLaunchKernelGGL(scale_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (size_t)n, dr, f);
    int32_t status = 0;
    int rc = apv_gevd_large(h, n, 1, dR, dR + nn, 0.0, nullptr, dU, dl, dr, mu, V, nullptr, dw, &status);     // jdiag(RB, RD, 'vector', true): no loading
    if (rc == APV_OK) {
        (void)hipMemcpyAsync(h_w, dw + (size_t)(V - 1) * n, sizeof(double) * n, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
    }
    if (rc == APV_OK && status == 2)
        return apv_fail(h, APV_ERR_NO_CONVERGE, "vast_static: eigen-iteration did not converge (Jacobi sweep cap reached); w was written");
    return rc;
}

// state: "response<p>" [C][N], "target_response<z>" [M][N], "input_block" [2][N] (rings, logical order); "R<q>" [n][n]
// (AA, BB, AB, BA), "r" [2][n], "lambda" [2][n], "w" [2][V][n], "U<z>" [n][n] (columns = eigenvectors, descending),
// "stats<p>" [C][S], "target_stats<z>" [M][S] (rings, logical order); and what a bit-for-bit resume also needs:
// "input_history<g>" [P-1+H] (the reference's lfilter states, apvast.py:115-120, as the inputs they came from),
// "overlap<p>" [C][N], "target_overlap<z>" [M][N] (WOLA overlap buffers, apvast.py:132-137), "out_overlap" [n_out][N]
// (apvast.py:148-151), "filter_spectra" [n_out][K] c128 (apvast.py:394-403, 417-422)
static int bb_lookup(apv_handle* h, const char* name, double** d, size_t* count, int* rows, int* len, int* off) {
    apv_bb* s = h->bb;
    const std::string nm(name);
    *rows = 0;
    auto idx = [&](const char* pre, int maxv) -> int {
        const size_t pl = std::strlen(pre);
        if (nm.size() == pl + 1 && nm.compare(0, pl, pre) == 0 && nm[pl] >= '0' && nm[pl] < '0' + maxv) return nm[pl] - '0';
        return -1;
    };
    int q;
    if ((q = idx("response", 4)) >= 0) { *d = s->resp[q]; *count = (size_t)s->C * s->N; *rows = s->C; *len = s->N; *off = s->ring_off; return APV_OK; }
    if ((q = idx("target_response", 2)) >= 0) { *d = s->tresp[q]; *count = (size_t)s->M * s->N; *rows = s->M; *len = s->N; *off = s->ring_off; return APV_OK; }
    if ((q = idx("stats", 4)) >= 0) { *d = s->stats[q]; *count = (size_t)s->C * s->S; *rows = s->C; *len = s->S; *off = s->stat_off; return APV_OK; }
    if ((q = idx("target_stats", 2)) >= 0) { *d = s->tstats[q]; *count = (size_t)s->M * s->S; *rows = s->M; *len = s->S; *off = s->stat_off; return APV_OK; }
    if (nm == "input_block") { *d = s->inblk; *count = (size_t)2 * s->N; *rows = 2; *len = s->N; *off = s->ring_off; return APV_OK; }
    if ((q = idx("input_history", 2)) >= 0) { *d = s->xhist[s->cur][q]; *count = (size_t)s->P - 1 + s->H; return APV_OK; }
    if ((q = idx("overlap", 4)) >= 0) { *d = s->ov[q]; *count = (size_t)s->C * s->N; return APV_OK; }
    if ((q = idx("target_overlap", 2)) >= 0) { *d = s->tov[q]; *count = (size_t)s->M * s->N; return APV_OK; }
    if (nm == "out_overlap") { *d = s->outov; *count = (size_t)s->n_out * s->N; return APV_OK; }
    if (nm == "filter_spectra") { *d = s->fspec; *count = (size_t)s->n_out * s->K * 2; return APV_OK; }
    if ((q = idx("U", 2)) >= 0) { *d = s->U + (size_t)q * s->n * s->n; *count = (size_t)s->n * s->n; return APV_OK; }
    if ((q = idx("R", 4)) >= 0) { *d = s->R + (size_t)q * s->n * s->n; *count = (size_t)s->n * s->n; return APV_OK; }
    if (nm == "r") { *d = s->r; *count = (size_t)2 * s->n; return APV_OK; }
    if (nm == "lambda") { *d = s->lam; *count = (size_t)2 * s->n; return APV_OK; }
    if (nm == "w") { *d = s->w; *count = (size_t)2 * s->V * s->n; return APV_OK; }
    if (nm == "input_spectrum") { *d = s->inspec; *count = (size_t)2 * s->K * 2; return APV_OK; }
    if ((q = idx("weights", 2)) >= 0 && s->nch > 0) { *d = s->Wgt[q]; *count = (size_t)s->M * s->K; return APV_OK; }
    return apv_fail(h, APV_ERR_STATE, std::string("unknown broadband state name: ") + name);
}

// The per-hop path solves for the eigenpairs the filters use (kernels_gevd_lead.hip).  lambda_* / U_* in full (apvast.py:380-387)
// are attributes somebody may read afterwards: the first read after such a hop runs the complete joint diagonalisation on
// the hop's matrices, which are still there (R is never modified by the solver).
static int bb_complete_eigenpairs(apv_handle* h) {
    apv_bb* s = h->bb;
    if (s->full_valid) return APV_OK;
    BCHK(h, hipSetDevice(h->device));
    const bool runA = s->zones & 1, runB = s->zones & 2;
    const int first = runA ? 0 : 1, batch = (runA && runB) ? 2 : 1, n = s->n;
    const size_t nn = (size_t)n * n;
    int32_t status[2] = {0, 0};
    h->gl_lead_rank = 0;
    const int rc = apv_gevd_large(h, n, batch, s->R + first * nn, s->R + (2 + first) * nn, s->rel_loading ? 0.0 : h->cfg.reg_dark,
                                  s->rel_dark_py ? s->nrm + 2 + first : nullptr, s->U + first * nn, s->lam + (size_t)first * n,
                                  nullptr, h->cfg.mu, 0, nullptr, nullptr, status);
    if (rc != APV_OK) return rc;
    if (status[0] == 2 || status[1] == 2)
        return apv_fail(h, APV_ERR_NO_CONVERGE, "eigen-iteration did not converge (Jacobi sweep cap reached) while completing lambda / U");
    s->full_valid = 1;
    return APV_OK;
}

int apv_bb_get_state(apv_handle* h, const char* name, double* h_dst, size_t count) {
    if (!h || !h->bb || !name || !h_dst) return apv_fail(h, APV_ERR_ARG, "null argument / broadband stream not initialised");
    double* d; size_t need; int rows, len, off;
    int rc = bb_lookup(h, name, &d, &need, &rows, &len, &off);
    if (rc != APV_OK) return rc;
    if (!std::strcmp(name, "lambda") || !std::strcmp(name, "U0") || !std::strcmp(name, "U1")) {
        if ((rc = bb_complete_eigenpairs(h)) != APV_OK) return rc;
    }
    if (count != need) return apv_fail(h, APV_ERR_STATE, "state size mismatch");
    BCHK(h, hipSetDevice(h->device));
    if (rows == 0) {
        BCHK(h, hipMemcpyAsync(h_dst, d, sizeof(double) * need, hipMemcpyDeviceToHost, h->stream));
        BCHK(h, hipStreamSynchronize(h->stream));
        return APV_OK;
    }
    std::vector<double> tmp(need);
    BCHK(h, hipMemcpyAsync(tmp.data(), d, sizeof(double) * need, hipMemcpyDeviceToHost, h->stream));
    BCHK(h, hipStreamSynchronize(h->stream));
    for (int r = 0; r < rows; ++r)
        for (int t = 0; t < len; ++t) h_dst[(size_t)r * len + t] = tmp[(size_t)r * len + (t + off) % len];
    return APV_OK;
}

int apv_bb_set_state(apv_handle* h, const char* name, const double* h_src, size_t count) {
    if (!h || !h->bb || !name || !h_src) return apv_fail(h, APV_ERR_ARG, "null argument / broadband stream not initialised");
    double* d; size_t need; int rows, len, off;
    int rc = bb_lookup(h, name, &d, &need, &rows, &len, &off);
    if (rc != APV_OK) return rc;
    if (count != need) return apv_fail(h, APV_ERR_STATE, "state size mismatch");
    BCHK(h, hipSetDevice(h->device));
    if (rows == 0) {
        BCHK(h, hipMemcpyAsync(d, h_src, sizeof(double) * need, hipMemcpyHostToDevice, h->stream));
        BCHK(h, hipStreamSynchronize(h->stream));
        return APV_OK;
    }
    std::vector<double> tmp(need);
    for (int r = 0; r < rows; ++r)
        for (int t = 0; t < len; ++t) tmp[(size_t)r * len + (t + off) % len] = h_src[(size_t)r * len + t];
    BCHK(h, hipMemcpyAsync(d, tmp.data(), sizeof(double) * need, hipMemcpyHostToDevice, h->stream));
    BCHK(h, hipStreamSynchronize(h->stream));               // tmp goes out of scope
    return APV_OK;
}

}  // extern "C"
