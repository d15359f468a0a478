// Streaming glue kernels around the per-bin update.
//   fir_hop_kernel      K1: RIR convolution of one hop to every control point   reference Python/apvast.py:167-194
//   apply_filters_kernel K3: output spectra = input spectrum x filter spectra    apvast.py:445-452
#include "apv_internal.h"

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
        const int mask = N - 1;
#pragma unroll
        for (int t = 0; t < FIR_TN; ++t)
            if (n0 + t < H) dst[(N - H + n0 + t + ring_off) & mask] = acc[t];
    }
}

// new_hist = [old_hist[H : H+P-1], x[0:H], zero pad]
__global__ void __launch_bounds__(256) hist_update_kernel(int P, int H, int pad, const float* __restrict__ old_hist,
                                                          const float* __restrict__ x, float* __restrict__ new_hist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int keep = P - 1;
    if (i < keep) new_hist[i] = old_hist[i + H];
    else if (i < keep + H) new_hist[i] = x[i - keep];
    else if (i < keep + H + pad) new_hist[i] = 0.f;
}

// ring[(N-H+n + ring_off) & (N-1)] = x[n]
__global__ void __launch_bounds__(256) ring_append_kernel(int N, int H, int ring_off, const float* __restrict__ x,
                                                          float* __restrict__ ring) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n < H) ring[(N - H + n + ring_off) & (N - 1)] = x[n];
}

template <typename W>
__global__ void __launch_bounds__(256) apply_filters_kernel(int K, int n_filt, int n_tgt,
                                                            const float2* __restrict__ in_spec,
                                                            const W* __restrict__ w, const float2* __restrict__ tgt,
                                                            float2* __restrict__ out) {
    // grid: x over k, y over channels (tiles of 16 channels x 16 bins so both sides stay reasonably coalesced)
    __shared__ float2 tile[16][17];
    const int k0 = blockIdx.x * 16, c0 = blockIdx.y * 16;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int n_ch = n_filt + n_tgt;
    {   // load: channel fastest (bin-major filter bank)
        const int k = k0 + ty, ch = c0 + tx;
        float2 f = make_float2(0.f, 0.f);
        if (k < K && ch < n_ch) {
            if (ch < n_filt) {
                const W v = w[(size_t)k * n_filt + ch];
                f = make_float2((float)v.x, (float)v.y);
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
            const float2 f = tile[tx][ty];
            const float2 x = in_spec[k];
            out[(size_t)ch * K + k] = make_float2(x.x * f.x - x.y * f.y, x.x * f.y + x.y * f.x);
        }
    }
}

}  // namespace

hipError_t apv_launch_fir_hop(int C, int P, int H, int N, int ring_off, const float* rir, const float* xhist,
                              float* resp, hipStream_t s) {
    if (C <= 0 || H <= 0) return hipSuccess;
    dim3 grid((C + 63) / 64, (H + FIR_TN - 1) / FIR_TN);
    hipLaunchKernelGGL(fir_hop_kernel, grid, dim3(64), 0, s, C, P, H, N, ring_off & (N - 1), rir, xhist, resp);
    return hipGetLastError();
}

hipError_t apv_launch_hist_update(int P, int H, int pad, const float* old_hist, const float* x, float* new_hist,
                                  hipStream_t s) {
    const int total = P - 1 + H + pad;
    hipLaunchKernelGGL(hist_update_kernel, dim3((total + 255) / 256), dim3(256), 0, s, P, H, pad, old_hist, x, new_hist);
    return hipGetLastError();
}

hipError_t apv_launch_ring_append(int N, int H, int ring_off, const float* x, float* ring, hipStream_t s) {
    hipLaunchKernelGGL(ring_append_kernel, dim3((H + 255) / 256), dim3(256), 0, s, N, H, ring_off & (N - 1), x, ring);
    return hipGetLastError();
}

int apv_fir_pad() { return FIR_TN; }

hipError_t apv_launch_apply_filters(int K, int n_filt, int n_tgt, const float2* in_spec, const void* w,
                                    int w_c128, const float2* tgt, float2* out, hipStream_t s) {
    const int n_ch = n_filt + n_tgt;
    if (n_ch <= 0 || K <= 0) return hipSuccess;
    dim3 grid((K + 15) / 16, (n_ch + 15) / 16);
    if (w_c128)
        hipLaunchKernelGGL(apply_filters_kernel<double2>, grid, dim3(256), 0, s, K, n_filt, n_tgt, in_spec,
                           (const double2*)w, tgt, out);
    else
        hipLaunchKernelGGL(apply_filters_kernel<float2>, grid, dim3(256), 0, s, K, n_filt, n_tgt, in_spec,
                           (const float2*)w, tgt, out);
    return hipGetLastError();
}
