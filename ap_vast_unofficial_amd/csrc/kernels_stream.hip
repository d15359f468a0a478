// Streaming glue kernels around the per-bin update.
//   fir_hop_kernel      K1: RIR convolution of one hop to every control point   reference Python/apvast.py:167-194
//   apply_filters_kernel K3: output spectra = input spectrum x filter spectra    apvast.py:445-452
#include "apv_internal.h"

#include <algorithm>

namespace {

constexpr int FIR_TN = 32;     // output samples per thread

// One thread = one control-point channel c, FIR_TN consecutive output samples.  The input history is
// wave-uniform (scalar loads); the taps are read coalesced across channels ([P][C], channel fastest).
__global__ void __launch_bounds__(64) fir_hop_kernel(int C, int P, int H, int N, int ring_off,
                                                     const float* __restrict__ rir, const float* __restrict__ xhist,
                                                     float* __restrict__ resp) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int n0 = blockIdx.y * FIR_TN;
    float acc[FIR_TN];
#pragma unroll
    for (int t = 0; t < FIR_TN; ++t) acc[t] = 0.f;
    const bool live = c < C;
    const float* xs = xhist + (P - 1) + n0;            // xs[t - p] = x[n0 + t - p]
    for (int p = 0; p < P; ++p) {
        const float r = live ? rir[(size_t)p * C + c] : 0.f;
#pragma unroll
        for (int t = 0; t < FIR_TN; ++t) acc[t] = __builtin_fmaf(r, xs[t - p], acc[t]);
    }
    if (live) {
        float* dst = resp + (size_t)c * N;
#pragma unroll
        for (int t = 0; t < FIR_TN; ++t)
            if (n0 + t < H) dst[(N - H + n0 + t + ring_off) % N] = acc[t];
    }
}

// ---- K1 on the matrix cores -------------------------------------------------------------------
// Y[n][c] = sum_p x[n-p] R[p][c] is the GEMM  T (H x P, Toeplitz, T[n][p] = x[n-p])  times  R (P x C).
// T is never materialised: the A operand of v_mfma_f32_32x32x2_f32 (lane l: A[i=l&31][k=l>>5]) is read
// straight from the input history in LDS at offset (n0 + i) - (p0 + k).  One wave = 32 samples x 32
// channels, one workgroup = 128 samples x 32 channels; blockIdx.z walks a small job table so that all
// paths (A->A, A->B, B->A, B->B, two targets) go out in ONE launch.
using f16v = __attribute__((ext_vector_type(16))) float;

__global__ void __launch_bounds__(256) fir_mfma_kernel(FirJobs jobs, int P, int H, int N, int ring_off) {
    extern __shared__ float xs[];                  // [P - 1 + 128] history window, then reused as 4 x [32][33] tiles
    const FirJob job = jobs.j[blockIdx.z];
    const int c0 = blockIdx.y * 32;
    if (c0 >= job.C) return;                       // uniform per workgroup
    const int C = job.C;
    const int n0 = blockIdx.x * 128;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int span = P - 1 + 128;
    for (int i = tid; i < span; i += 256) xs[i] = job.xh[n0 + i];      // xh is padded to P-1+ceil(H/128)*128
    __syncthreads();
    const int i = lane & 31, kk = lane >> 5, nb = 32 * wave;
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // channels past C read a valid (clamped) column; their results are never stored
    const int cc = (c0 + i) < C ? (c0 + i) : (C - 1);
    const float* bptr = job.rir + (size_t)kk * C + cc;
    const float* aptr = xs + (P - 1) + nb + i - kk;
    const int Peven = P & ~1;
    // software pipeline: the taps of the next group of 8 k-steps are in flight while this group's MFMAs issue
    constexpr int G = 8;
    const int ngroups = Peven / (2 * G);
    float bn[G];
#pragma unroll
    for (int u = 0; u < G; ++u) bn[u] = (ngroups > 0) ? bptr[(size_t)(2 * u) * C] : 0.f;
    for (int g = 0; g < ngroups; ++g) {
        const int p0 = g * 2 * G;
        float a[G], b[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            b[u] = bn[u];
            a[u] = aptr[-(p0 + 2 * u)];
        }
        if (g + 1 < ngroups) {
#pragma unroll
            for (int u = 0; u < G; ++u) bn[u] = bptr[(size_t)(p0 + 2 * G + 2 * u) * C];
        }
#pragma unroll
        for (int u = 0; u < G; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
    for (int p0 = ngroups * 2 * G; p0 < Peven; p0 += 2) {
        const float a = aptr[-p0];
        const float b = bptr[(size_t)p0 * C];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (P & 1) {                                   // last odd tap: only k = 0 carries data
        const float a = (kk == 0) ? aptr[-Peven] : 0.f;
        const float b = (kk == 0) ? bptr[(size_t)Peven * C] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();                               // everyone is done with the history window
    // accumulator (row = sample, col = channel): row = (r&3) + 8(r>>2) + 4(lane>>5), col = lane&31
    float* tile = xs + wave * (32 * 33);           // [channel][sample], padded
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
        tile[i * 33 + row] = acc[r];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int c = (lane >> 3) + 8 * t, q = (lane & 7) * 4;
        const int n = n0 + nb + q;
        if (c0 + c < C && n < H) {
            float* dst = job.resp + (size_t)(c0 + c) * N;
            const int base = (N - H + n + ring_off) % N;
            if (((base & 3) == 0) && (n + 3 < H) && (base + 3 < N) && ((N & 3) == 0)) {
                *reinterpret_cast<float4*>(dst + base) =
                    make_float4(tile[c * 33 + q], tile[c * 33 + q + 1], tile[c * 33 + q + 2], tile[c * 33 + q + 3]);
            } else {
                for (int u = 0; u < 4; ++u)
                    if (n + u < H) dst[(base + u) % N] = tile[c * 33 + q + u];
            }
        }
    }
}

// both input signals of a hop in one launch (blockIdx.y = signal): new history = [old[H:], x, zeros(pad)] and the hop
// appended to the input-block ring
template <typename T>
struct InputUpdate {
    const T* old_hist[2];
    T* new_hist[2];
};
template <typename T>
__global__ void __launch_bounds__(256) input_update_kernel(int P, int H, int pad, int N, int ring_off, InputUpdate<T> u,
                                                           const T* __restrict__ xin, T* __restrict__ inblk) {
    const int g = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const T* x = xin + (size_t)g * H;
    const int keep = P - 1;
    if (i < keep) u.new_hist[g][i] = u.old_hist[g][i + H];
    else if (i < keep + H) u.new_hist[g][i] = x[i - keep];
    else if (i < keep + H + pad) u.new_hist[g][i] = (T)0;
    if (i < H) inblk[(size_t)g * N + (N - H + i + ring_off) % N] = x[i];
}

// ---- K1 in float64 on v_mfma_f64_16x16x4_f64 (both stream modes) ---------------------------------
//   y[n][c] = sum_p xh[P-1+n-p] rir[p][c]                                           apvast.py:171-192 (lfilter)
// A workgroup owns NT x 16 samples x 16 channels; its four waves split the taps, each streaming its share of the RIR
// slab straight into MFMA B operands (one load serves all NT sample tiles) while the A operands are windows of the
// input history held in LDS; the four partial tiles are summed through LDS.  All paths and targets of the hop are one
// launch (job table).
using d4 = __attribute__((ext_vector_type(4))) double;

template <int NT>
__global__ void __launch_bounds__(256) fir_f64_mfma_kernel(int P, int H, int N, int ring_off, int njobs, FirJobsD jobs) {
    extern __shared__ double fir_lds[];          // [P - 1 + 16 NT] history window; afterwards [4][NT][264] partial tiles
    // Workgroups go to the eight XCDs round-robin.  All sample tiles of a channel tile read the same P x 16 taps (102 KB at
    // P = 800; the six filter banks together are 7 MB, more than one XCD's L2): workgroup b takes the b/8-th (channel tile,
    // sample tile) pair, sample tile fastest, of the (b mod 8)-th eighth of the launch, so the pairs that share taps run
    // back to back on ONE XCD and the taps come from HBM once.
    const int lin = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y, total = (int)(gridDim.x * gridDim.y);
    const int q8 = total >> 3, r8 = total & 7, g8 = lin & 7;
    const int wg = g8 * q8 + min(g8, r8) + (lin >> 3);
    const int by = wg / (int)gridDim.x, bx = wg - by * (int)gridDim.x;
    int j = 0;
    while (j + 1 < njobs && by >= jobs.tile0[j + 1]) ++j;
    const int C = jobs.C[j];
    const double* __restrict__ rir = jobs.rir[j];
    const double* __restrict__ xh = jobs.xh[j];
    const int c0 = (by - jobs.tile0[j]) * 16, n0 = bx * 16 * NT;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, il = lane & 15, kq = lane >> 4;
    constexpr int SPAN = 16 * NT - 1;
    double* xw = fir_lds;
    double* part = fir_lds;                      // the partial tiles take the window's place once every wave is done with it
    for (int i = tid; i < P + SPAN; i += 256) xw[i] = xh[n0 + i];
    const int steps_total = (P + 3) >> 2, spw = (steps_total + 3) >> 2;
    const int s_begin = wave * spw, s_end = min(s_begin + spw, steps_total);
    const int c = c0 + il;
    const bool c_ok = c < C;
    __syncthreads();
    d4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (d4){0, 0, 0, 0};
    constexpr int G = NT > 1 ? 8 : 32;           // k-steps whose taps are in flight together
    // taps of the NEXT group are requested before the MFMAs of the current one are issued: a tap comes from L2 or the
    // Infinity Cache (the six filter banks are 7 MB) and its latency is longer than one group's 8 NT MFMAs
    auto load_taps = [&](double (&dst)[G], int s0) {
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const int pt = 4 * (s0 + q) + kq;
            dst[q] = (s0 + q < s_end && pt < P && c_ok) ? rir[(size_t)pt * C + c] : 0.0;
        }
    };
    double bv[G], bn[G];
    load_taps(bv, s_begin);
    for (int s0 = s_begin; s0 < s_end; s0 += G) {
        load_taps(bn, s0 + G);                               // past s_end: all zeros, never used
#pragma unroll
        for (int q = 0; q < G; ++q) {
            if (s0 + q >= s_end) break;                      // wave-uniform: the last group of a wave is usually partial
            const int pt = 4 * (s0 + q) + kq;
            const int wi = P - 1 + il - pt;                  // taps past P carry a zero B operand
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int wj = wi + 16 * t;
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xw[wj > 0 ? wj : 0], bv[q], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < G; ++q) bv[q] = bn[q];
    }
    // partial tiles -> LDS (row stride RS keeps the transposed read below off a single bank), then each thread sums the four
    // waves for NT consecutive samples of one channel: the ring is written in runs of 16 NT samples per channel
    constexpr int RS = 66, TS = 4 * RS;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(wave * NT + t) * TS + r * RS + lane] = acc[t][r];
    __syncthreads();
    // accumulator element (r, lane) of sample tile t: sample 16 t + (lane >> 4) + 4 r, channel lane & 15
    const int ch = tid >> 4, qs = tid & 15;
    const bool ch_ok = c0 + ch < C;
    double* __restrict__ dst = jobs.resp[j] + (size_t)(c0 + ch) * N;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int sm = qs * NT + u, t = sm >> 4, wi = sm & 15;
        const int e = t * TS + (wi >> 2) * RS + (wi & 3) * 16 + ch;
        const double v = part[e] + part[NT * TS + e] + part[2 * NT * TS + e] + part[3 * NT * TS + e];
        const int n = n0 + sm;
        if (n < H && ch_ok) dst[(N - H + n + ring_off) % N] = v;
    }
}

// ---- perceptual weighting (van de Par 2005, Matlab/ControlMethods/perceptualModel.m:118-139, 177-190) --------
// One workgroup = one control point m of one zone.  spec: target spectra, bin-major [K][M] c64 (unscaled rfft);
// G2 [K][nch] and G2T [nch][K]: squared outer/middle-ear x gammatone responses.  Out: W [K][M] real weights.
//   masker_i = sum_k G2[k][i] |sqrt(2)/N S[k]|^2;  w^2[k] = Cs Leff sum_i G2[k][i] / (masker_i + Ca)
//   W = w / ||w||  over the K bins (norm_mode 0, apvast.py:322-324) or over the full symmetric curve (1)
// spec element (k, m) at spec[k * s_k + m * s_m]; W element (k, m) at W[k * w_k + m * w_m]
template <typename S2, typename WT>
__global__ void __launch_bounds__(256) perceptual_weights_kernel(int K, int M, int nch, const S2* __restrict__ spec,
                                                                 long s_k, long s_m, const double* __restrict__ G2,
                                                                 const double* __restrict__ G2T, double Cs, double Ca,
                                                                 double Leff, double fscale2, int norm_mode,
                                                                 WT* __restrict__ W, long w_k, long w_m) {
    extern __shared__ double sm[];
    double* P2 = sm;               // [K]
    double* inv = sm + K;          // [nch]
    double* red = inv + nch;       // [256]
    const int m = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < K; k += 256) {
        const S2 v = spec[(size_t)k * s_k + (size_t)m * s_m];
        P2[k] = fscale2 * ((double)v.x * v.x + (double)v.y * v.y);
    }
    __syncthreads();
    for (int i = 0; i < nch; ++i) {
        double a = 0.0;
        const double* g = G2T + (size_t)i * K;
        for (int k = tid; k < K; k += 256) a += g[k] * P2[k];
        red[tid] = a;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (tid < w) red[tid] += red[tid + w];
            __syncthreads();
        }
        if (tid == 0) inv[i] = 1.0 / (red[0] + Ca);
        __syncthreads();
    }
    double part = 0.0;
    for (int k = tid; k < K; k += 256) {
        double a = 0.0;
        const double* g = G2 + (size_t)k * nch;
        for (int i = 0; i < nch; ++i) a += g[i] * inv[i];
        a *= Cs * Leff;
        P2[k] = a;                                                  // w^2[k]
        part += (norm_mode == 1 && k > 0 && k < K - 1) ? 2.0 * a : a;
    }
    red[tid] = part;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    const double inorm = 1.0 / red[0];
    for (int k = tid; k < K; k += 256) W[(size_t)k * w_k + (size_t)m * w_m] = (WT)sqrt(P2[k] * inorm);
}

// spec[k][c] *= W[k][c / L]   (bin-major c64; L = 1 scales the target spectra themselves)
template <typename S2, typename WT>
__global__ void __launch_bounds__(256) scale_spectra_kernel(int K, int C, int L, S2* __restrict__ spec,
                                                            const WT* __restrict__ W) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)K * C) return;
    const int k = (int)(idx / C), c = (int)(idx - (size_t)k * C);
    const WT w = W[(size_t)k * (C / L) + c / L];
    S2 v = spec[idx];
    v.x *= w;
    v.y *= w;
    spec[idx] = v;
}

// channel-major float64 form (broadband mode): spec[c][k] *= W[c / L][k]
__global__ void __launch_bounds__(256) scale_spectra_cm_f64_kernel(int K, int C, int L, double2* __restrict__ spec,
                                                                   const double* __restrict__ W) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)K * C) return;
    const int c = (int)(idx / K), k = (int)(idx - (size_t)c * K);
    const double w = W[(size_t)(c / L) * K + k];
    double2 v = spec[idx];
    v.x *= w;
    v.y *= w;
    spec[idx] = v;
}

// Output spectra of a hop in one launch: job j multiplies the input spectrum in[j] with n_filt[j] filters of the bin-major
// bank w[j] (c64 or c128) and/or n_tgt[j] channel-major target filters tgt[j]; blockIdx.y walks the 16-channel tiles of
// all jobs.
constexpr int APPLY_MAX_JOBS = 4;
struct ApplyJobs {
    const void* in[APPLY_MAX_JOBS];
    const void* w[APPLY_MAX_JOBS];
    const void* tgt[APPLY_MAX_JOBS];
    void* out[APPLY_MAX_JOBS];
    int n_filt[APPLY_MAX_JOBS], n_tgt[APPLY_MAX_JOBS], tile0[APPLY_MAX_JOBS + 1];
    int n;
};
// W: element type of the filter bank (float2 | double2); S2: element type of the spectra (float2 | double2)
template <typename W, typename S2>
__global__ void __launch_bounds__(256) apply_filters_kernel(int K, ApplyJobs jobs) {
    // grid: x over k, y over channels (tiles of 16 channels x 16 bins so both sides stay reasonably coalesced)
    using R = decltype(S2::x);
    __shared__ S2 tile[16][17];
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.y >= jobs.tile0[j + 1]) ++j;
    const int n_filt = jobs.n_filt[j], n_tgt = jobs.n_tgt[j];
    const W* __restrict__ w = reinterpret_cast<const W*>(jobs.w[j]);
    const S2* __restrict__ tgt = reinterpret_cast<const S2*>(jobs.tgt[j]);
    const S2* __restrict__ in_spec = reinterpret_cast<const S2*>(jobs.in[j]);
    S2* __restrict__ out = reinterpret_cast<S2*>(jobs.out[j]);
    const int k0 = blockIdx.x * 16, c0 = ((int)blockIdx.y - jobs.tile0[j]) * 16;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int n_ch = n_filt + n_tgt;
    {   // load: channel fastest (bin-major filter bank)
        const int k = k0 + ty, ch = c0 + tx;
        S2 f;
        f.x = (R)0;
        f.y = (R)0;
        if (k < K && ch < n_ch) {
            if (ch < n_filt) {
                const W v = w[(size_t)k * n_filt + ch];
                f.x = (R)v.x;
                f.y = (R)v.y;
            } else {
                f = tgt[(size_t)(ch - n_filt) * K + k];
            }
        }
        tile[ty][tx] = f;
    }
    __syncthreads();
    {   // store: bin fastest (channel-major spectra for the synthesis kernel)
        const int k = k0 + tx, ch = c0 + ty;
        if (k < K && ch < n_ch) {
            const S2 f = tile[tx][ty];
            const S2 x = in_spec[k];
            S2 y;
            y.x = x.x * f.x - x.y * f.y;
            y.y = x.x * f.y + x.y * f.x;
            out[(size_t)ch * K + k] = y;
        }
    }
}

}  // namespace

hipError_t apv_launch_fir_hop(int C, int P, int H, int N, int ring_off, const float* rir, const float* xhist,
                              float* resp, hipStream_t s) {
    if (C <= 0 || H <= 0) return hipSuccess;
    dim3 grid((C + 63) / 64, (H + FIR_TN - 1) / FIR_TN);
    hipLaunchKernelGGL(fir_hop_kernel, grid, dim3(64), 0, s, C, P, H, N, ring_off % N, rir, xhist, resp);
    return hipGetLastError();
}

// f64 = 0: c64 bin-major spectra [K][M] -> float weights [K][M]; f64 = 1: c128 spectra -> double weights
hipError_t apv_launch_perceptual_weights(int f64, int K, int M, int nch, const void* spec, const double* G2, const double* G2T,
                                         double Cs, double Ca, double Leff, int N, int norm_mode, void* W, hipStream_t s) {
    const size_t lds = sizeof(double) * ((size_t)K + nch + 256);
    const double fs2 = 2.0 / ((double)N * (double)N);
    if (f64)
        hipLaunchKernelGGL((perceptual_weights_kernel<double2, double>), dim3(M), dim3(256), lds, s, K, M, nch, (const double2*)spec,
                           (long)M, 1L, G2, G2T, Cs, Ca, Leff, fs2, norm_mode, (double*)W, (long)M, 1L);
    else
        hipLaunchKernelGGL((perceptual_weights_kernel<float2, float>), dim3(M), dim3(256), lds, s, K, M, nch, (const float2*)spec,
                           (long)M, 1L, G2, G2T, Cs, Ca, Leff, fs2, norm_mode, (float*)W, (long)M, 1L);
    return hipGetLastError();
}

// float64, channel-major spectra [M][K] -> weights [M][K] (broadband mode)
hipError_t apv_launch_perceptual_weights_f64(int K, int M, int nch, const double2* spec, const double* G2,
                                             const double* G2T, double Cs, double Ca, double Leff, int N, int norm_mode,
                                             double* W, hipStream_t s) {
    const size_t lds = sizeof(double) * ((size_t)K + nch + 256);
    hipLaunchKernelGGL((perceptual_weights_kernel<double2, double>), dim3(M), dim3(256), lds, s, K, M, nch, spec, 1L, (long)K,
                       G2, G2T, Cs, Ca, Leff, 2.0 / ((double)N * (double)N), norm_mode, W, 1L, (long)K);
    return hipGetLastError();
}

hipError_t apv_launch_scale_spectra_cm_f64(int K, int C, int L, double2* spec, const double* W, hipStream_t s) {
    const size_t total = (size_t)K * C;
    hipLaunchKernelGGL(scale_spectra_cm_f64_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, K, C, L, spec, W);
    return hipGetLastError();
}

// out[k][c] = in[(k / g) g C + c g + k % g]: grouped spectra [K / g][C][g] back to bin-major [K][C] (attribute reads only)
template <typename E>
__global__ void __launch_bounds__(256) ungroup_spectra_kernel(int K, int C, int g, const E* __restrict__ in, E* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)K * C) return;
    const int k = (int)(idx / C), c = (int)(idx % C);
    out[idx] = in[(size_t)(k / g) * g * C + (size_t)c * g + (k % g)];
}
hipError_t apv_launch_ungroup_spectra(int f64, int K, int C, int g, const void* in, void* out, hipStream_t s) {
    const unsigned blocks = (unsigned)(((size_t)K * C + 255) / 256);
    if (f64) hipLaunchKernelGGL(ungroup_spectra_kernel<double2>, dim3(blocks), dim3(256), 0, s, K, C, g, (const double2*)in, (double2*)out);
    else hipLaunchKernelGGL(ungroup_spectra_kernel<float2>, dim3(blocks), dim3(256), 0, s, K, C, g, (const float2*)in, (float2*)out);
    return hipGetLastError();
}

hipError_t apv_launch_scale_spectra(int f64, int K, int C, int L, void* spec, const void* W, hipStream_t s) {
    const size_t total = (size_t)K * C;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (f64) hipLaunchKernelGGL((scale_spectra_kernel<double2, double>), grid, dim3(256), 0, s, K, C, L, (double2*)spec, (const double*)W);
    else hipLaunchKernelGGL((scale_spectra_kernel<float2, float>), grid, dim3(256), 0, s, K, C, L, (float2*)spec, (const float*)W);
    return hipGetLastError();
}

hipError_t apv_launch_input_update(int f64, int P, int H, int pad, int N, int ring_off, const void* const old_hist[2],
                                   void* const new_hist[2], const void* xin, void* inblk, hipStream_t s) {
    const int total = P - 1 + H + pad;
    const dim3 grid((total + 255) / 256, 2);
    if (f64) {
        InputUpdate<double> u{{(const double*)old_hist[0], (const double*)old_hist[1]}, {(double*)new_hist[0], (double*)new_hist[1]}};
        hipLaunchKernelGGL(input_update_kernel<double>, grid, dim3(256), 0, s, P, H, pad, N, ring_off % N, u, (const double*)xin, (double*)inblk);
    } else {
        InputUpdate<float> u{{(const float*)old_hist[0], (const float*)old_hist[1]}, {(float*)new_hist[0], (float*)new_hist[1]}};
        hipLaunchKernelGGL(input_update_kernel<float>, grid, dim3(256), 0, s, P, H, pad, N, ring_off % N, u, (const float*)xin, (float*)inblk);
    }
    return hipGetLastError();
}

// ---- whole-signal path: a chunk of hops at a time (stream.hip, process_signal_chunked_t) -------------------------------------
// rows of `len` samples between buffers with their own row strides and ring positions:
//   dst[r ds + (d0 + m) mod dmod] = src[r ss + (s0 + m) mod smod],  m < len
template <typename T>
__global__ void __launch_bounds__(256) rows_copy_kernel(T* __restrict__ dst, long ds, int d0, int dmod, const T* __restrict__ src,
                                                        long ss, int s0, int smod, int len) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= len) return;
    const size_t r = blockIdx.y;
    dst[r * ds + (d0 + m) % dmod] = src[r * ss + (s0 + m) % smod];
}

// The input side of a chunk of nc hops (pinned staging pin [nc][2][H]) in one launch: the hops appended to the linear
// input-block buffers inL [2][RL] at positions N - H + i H ..., and the input histories as they are AFTER the chunk, i.e. what nc
// input updates (input_update_kernel) would leave: the last P - 1 + H samples of [old history's last P - 1 | hop 0 | hop 1 ...]
// (length P - 1 + nc H), then `pad` zeros.
template <typename T>
__global__ void __launch_bounds__(256) chunk_inputs_kernel(int P, int H, int N, int nc, int pad, int RL, InputUpdate<T> u,
                                                           const T* __restrict__ pin, T* __restrict__ inL) {
    const int g = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int keep = P - 1;
    if (i < nc * H) {
        const int q = i / H, t = i - q * H;
        inL[(size_t)g * RL + (N - H) + i] = pin[((size_t)q * 2 + g) * H + t];
    }
    if (i < keep + H) {
        // with S = [old history's last P - 1 | hop 0 | hop 1 ...] the history hop j works from is S[j H ... j H + P - 1 + H): after
        // the chunk's last hop, j = nc - 1
        const int m = (nc - 1) * H + i;
        T v;
        if (m < keep) v = u.old_hist[g][H + m];
        else {
            const int q = (m - keep) / H, t = (m - keep) - q * H;
            v = pin[((size_t)q * 2 + g) * H + t];
        }
        u.new_hist[g][i] = v;
    } else if (i < keep + H + pad) {
        u.new_hist[g][i] = (T)0;
    }
}

hipError_t apv_launch_rows_copy(int f64, int rows, int len, void* dst, long ds, int d0, int dmod, const void* src, long ss, int s0,
                                int smod, hipStream_t s) {
    if (rows <= 0 || len <= 0) return hipSuccess;
    const dim3 grid((len + 255) / 256, rows);
    if (f64) hipLaunchKernelGGL(rows_copy_kernel<double>, grid, dim3(256), 0, s, (double*)dst, ds, d0, dmod, (const double*)src, ss, s0, smod, len);
    else hipLaunchKernelGGL(rows_copy_kernel<float>, grid, dim3(256), 0, s, (float*)dst, ds, d0, dmod, (const float*)src, ss, s0, smod, len);
    return hipGetLastError();
}

hipError_t apv_launch_chunk_inputs(int f64, int P, int H, int N, int nc, int pad, int RL, const void* const old_hist[2],
                                   void* const new_hist[2], const void* pin, void* inL, hipStream_t s) {
    const int span = (nc * H > P - 1 + H + pad) ? nc * H : P - 1 + H + pad;
    const dim3 grid((span + 255) / 256, 2);
    if (f64) {
        InputUpdate<double> u{{(const double*)old_hist[0], (const double*)old_hist[1]}, {(double*)new_hist[0], (double*)new_hist[1]}};
        hipLaunchKernelGGL(chunk_inputs_kernel<double>, grid, dim3(256), 0, s, P, H, N, nc, pad, RL, u, (const double*)pin, (double*)inL);
    } else {
        InputUpdate<float> u{{(const float*)old_hist[0], (const float*)old_hist[1]}, {(float*)new_hist[0], (float*)new_hist[1]}};
        hipLaunchKernelGGL(chunk_inputs_kernel<float>, grid, dim3(256), 0, s, P, H, N, nc, pad, RL, u, (const float*)pin, (float*)inL);
    }
    return hipGetLastError();
}

int apv_fir_pad() { return 128; }      // history buffers are padded so that a 128-sample tile never reads past the end

hipError_t apv_launch_fir_jobs(const FirJobs& jobs, int P, int H, int N, int ring_off, hipStream_t s) {
    if (jobs.n <= 0 || H <= 0) return hipSuccess;
    int maxC = 0;
    for (int j = 0; j < jobs.n; ++j) maxC = jobs.j[j].C > maxC ? jobs.j[j].C : maxC;
    dim3 grid((H + 127) / 128, (maxC + 31) / 32, jobs.n);
    size_t lds = sizeof(float) * (size_t)(P - 1 + 128);
    const size_t tiles = sizeof(float) * 4 * 32 * 33;
    if (lds < tiles) lds = tiles;
    hipLaunchKernelGGL(fir_mfma_kernel, grid, dim3(256), lds, s, jobs, P, H, N, ring_off % N);
    return hipGetLastError();
}

// float64 jobs: fills tile0 from C; the input histories must be readable up to P - 1 + roundup(H, 64) samples
// (zero tail: apv_fir_pad_f64())
int apv_fir_pad_f64() { return 64; }

hipError_t apv_launch_fir_jobs_f64(FirJobsD jobs, int njobs, int P, int H, int N, int ring_off, hipStream_t s) {
    if (njobs <= 0 || H <= 0) return hipSuccess;
    if (njobs > FIR_JOBS_D) return hipErrorInvalidValue;
    int tiles = 0;
    for (int j = 0; j < njobs; ++j) {
        jobs.tile0[j] = tiles;
        tiles += (jobs.C[j] + 15) / 16;
    }
    jobs.tile0[njobs] = tiles;
    if (tiles == 0) return hipSuccess;
    // four sample tiles per workgroup once the hop is long enough to still fill the chip: every tap load then feeds four MFMAs
    if (H >= 256) {
        constexpr int NT = 4;
        const size_t lds = sizeof(double) * std::max((size_t)P + 16 * NT, (size_t)4 * NT * 264);
        hipLaunchKernelGGL(fir_f64_mfma_kernel<NT>, dim3((H + 16 * NT - 1) / (16 * NT), tiles), dim3(256), lds, s, P, H, N,
                           ring_off % N, njobs, jobs);
    } else {
        const size_t lds = sizeof(double) * std::max((size_t)P + 16, (size_t)4 * 264);
        hipLaunchKernelGGL(fir_f64_mfma_kernel<1>, dim3((H + 15) / 16, tiles), dim3(256), lds, s, P, H, N, ring_off % N, njobs, jobs);
    }
    return hipGetLastError();
}

// spec_f64 = 0: c64 spectra (in, tgt, out), 1: c128; w_c128: element type of the bin-major filter banks
hipError_t apv_launch_apply_jobs(int K, int n_jobs, const void* const* in_spec, const void* const* w, const void* const* tgt,
                                 void* const* out, const int* n_filt, const int* n_tgt, int w_c128, int spec_f64, hipStream_t s) {
    if (n_jobs < 1 || n_jobs > APPLY_MAX_JOBS || K <= 0) return hipErrorInvalidValue;
    ApplyJobs jobs{};
    int tiles = 0;
    for (int j = 0; j < n_jobs; ++j) {
        jobs.in[j] = in_spec[j]; jobs.w[j] = w[j]; jobs.tgt[j] = tgt[j]; jobs.out[j] = out[j];
        jobs.n_filt[j] = n_filt[j]; jobs.n_tgt[j] = n_tgt[j];
        jobs.tile0[j] = tiles;
        tiles += (n_filt[j] + n_tgt[j] + 15) / 16;
    }
    jobs.tile0[n_jobs] = tiles;
    jobs.n = n_jobs;
    if (tiles == 0) return hipSuccess;
    const dim3 grid((K + 15) / 16, tiles);
    if (spec_f64) {
        if (w_c128) hipLaunchKernelGGL((apply_filters_kernel<double2, double2>), grid, dim3(256), 0, s, K, jobs);
        else hipLaunchKernelGGL((apply_filters_kernel<float2, double2>), grid, dim3(256), 0, s, K, jobs);
    } else {
        if (w_c128) hipLaunchKernelGGL((apply_filters_kernel<double2, float2>), grid, dim3(256), 0, s, K, jobs);
        else hipLaunchKernelGGL((apply_filters_kernel<float2, float2>), grid, dim3(256), 0, s, K, jobs);
    }
    return hipGetLastError();
}
